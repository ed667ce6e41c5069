"""CPU model of how the library's look-back kernels take their partitions (csrc/gs_sort.hip onesweep_kernel, csrc/gs_raster.hip bin_emit_kernel;
DESIGN.md section 4.1), searched for stalls under adversarial dispatch.

The machine: a fixed number of workgroup slots shared by the kernels of several streams.  A kernel is a persistent grid of G workgroups walking P > G
partitions; the dispatcher starts a kernel's workgroups in blockIdx order, one whenever a slot is free, choosing freely AMONG the kernels.  A workgroup
that holds partition p publishes p's totals after a finite amount of local work, then waits (spins, keeping its slot) until every partition before p has
published, finishes p, and takes its next partition; it exits -- freeing the slot -- when there is none left.  Three ways of taking partitions:

    one_counter    every partition from one counter: whoever runs claims the lowest unclaimed one                      (lanes, second contexts, shared GPUs)
    static_first   workgroup b takes partition b first, the counter hands out the rest                                  (a context alone on its GPU)
    per_class      C counters, workgroup b draws from counter b % C, ticket t of class c is partition t * C + c        (rounds 1-5)

Claims the kernels' comments and gs_context_set_shared_gpu's documentation rest on, checked here by random adversarial schedules:
  * a one_counter kernel finishes with ANY number of slots, alone or beside any number of its kind: a running workgroup always holds the lowest unfinished
    partition or waits on one that a running workgroup holds;
  * static_first and per_class need co-residency -- the whole grid (static_first) or one workgroup of every class (per_class) must be able to hold slots at
    the same time; the launch code sizes a persistent grid to what is resident at once, so ALONE they finish -- and with fewer slots than that they stall
    even alone (why the forms that assume it are reserved for a context that has the GPU to itself);
  * ONE static_first kernel beside any number of one_counter kernels finishes (those end with whatever workgroups they have and release their slots);
  * TWO static_first kernels can hold each other's slots for good although each would fit alone (what frames in flight met at C3 / C5 / C2d / C4 before
    the lanes got one_counter), and so can two per_class kernels once neither has a workgroup of every class running (what two processes on one GPU met)."""
import random

import pytest


class Kernel:
    def __init__(self, form, grid, parts, classes=2):
        self.form, self.G, self.P, self.C = form, grid, parts, classes
        self.dispatched = 0                       # workgroups started so far (blockIdx order)
        self.counter = [0] * (classes if form == "per_class" else 1)
        self.published = [False] * parts
        self.done = [False] * parts
        self.running = {}                         # blockIdx -> [partition or None, rounds taken]
        self.exited = 0

    def claim(self, b, rnd):
        if self.form == "one_counter":
            p = self.counter[0]; self.counter[0] += 1
        elif self.form == "static_first":
            if rnd == 0:
                p = b
            else:
                p = self.G + self.counter[0]; self.counter[0] += 1
        else:
            c = b % self.C
            p = self.counter[c] * self.C + c; self.counter[c] += 1
        return p if p < self.P else None

    def finished(self):
        return self.exited == self.G


def run(forms, slots, grid, parts, seed, classes=2, max_steps=20000):
    """True if every kernel finishes under this random schedule; False if the machine reaches a state in which nothing can move."""
    rng = random.Random(seed)
    ks = [Kernel(f, grid, parts, classes) for f in forms]
    free = slots
    for _ in range(max_steps):
        if all(k.finished() for k in ks):
            return True
        moves = []
        if free > 0:
            moves += [("dispatch", k) for k in ks if k.dispatched < k.G]
        for k in ks:
            for b, (p, rnd) in k.running.items():
                if p is None:
                    moves.append(("claim", k, b))
                elif not k.published[p]:
                    moves.append(("publish", k, b))
                elif all(k.published[q] for q in range(p)):
                    moves.append(("finish", k, b))
                # else: spinning on a predecessor that has not published -- no move of its own
        if not moves:
            return False                          # every running workgroup spins, no slot is free: held for good
        m = rng.choice(moves)
        if m[0] == "dispatch":
            k = m[1]; k.running[k.dispatched] = [None, 0]; k.dispatched += 1; free -= 1
        else:
            k, b = m[1], m[2]
            st = k.running[b]
            if m[0] == "claim":
                p = k.claim(b, st[1]); st[1] += 1
                if p is None:
                    del k.running[b]; k.exited += 1; free += 1
                else:
                    st[0] = p
            elif m[0] == "publish":
                k.published[st[0]] = True
            else:
                k.done[st[0]] = True; st[0] = None
    raise AssertionError("schedule did not terminate")


SEEDS = range(400)


@pytest.mark.parametrize("slots", [1, 2, 5])
def test_a_one_counter_kernel_needs_no_residency_at_all(slots):
    assert all(run(["one_counter"], slots, grid=4, parts=11, seed=s) for s in SEEDS)


@pytest.mark.parametrize("form,slots", [("static_first", 4), ("static_first", 6), ("per_class", 2), ("per_class", 4)])
def test_the_other_forms_finish_alone_when_the_grid_fits(form, slots):
    assert all(run([form], slots, grid=4, parts=11, seed=s, classes=2) for s in SEEDS)


@pytest.mark.parametrize("form,slots", [("static_first", 3), ("static_first", 1), ("per_class", 1)])
def test_the_other_forms_stall_alone_when_it_does_not(form, slots):
    assert not any(run([form], slots, grid=4, parts=11, seed=s, classes=2) for s in SEEDS)


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("slots", [1, 3])
def test_one_counter_kernels_finish_together(n, slots):
    assert all(run(["one_counter"] * n, slots=slots, grid=3, parts=9, seed=s) for s in SEEDS)


@pytest.mark.parametrize("others", [1, 2, 3])
def test_one_static_first_kernel_beside_one_counter_kernels_finishes(others):
    assert all(run(["static_first"] + ["one_counter"] * others, slots=3, grid=3, parts=9, seed=s) for s in SEEDS)
    assert all(run(["one_counter"] * others + ["static_first"], slots=4, grid=4, parts=10, seed=s) for s in SEEDS)


def test_two_static_first_kernels_can_hold_each_other():
    outcomes = [run(["static_first", "static_first"], slots=3, grid=3, parts=9, seed=s) for s in SEEDS]
    assert not all(outcomes) and any(outcomes)              # some schedules stall, some get through: what made it a rare, size-dependent hang


def test_two_per_class_kernels_can_hold_each_other():
    # two classes, two slots: each kernel alone gets one workgroup of either class; together, one workgroup each -- of the same class
    outcomes = [run(["per_class", "per_class"], slots=2, grid=4, parts=12, seed=s, classes=2) for s in SEEDS]
    assert not all(outcomes) and any(outcomes)
