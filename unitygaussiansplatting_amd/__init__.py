"""unitygaussiansplatting_amd -- MI355X-native drop-in for the per-frame render path of
aras-p/UnityGaussianSplatting (sort keys -> Onesweep depth sort -> per-splat view data ->
tile-binned front-to-back composite -> resolve), behind the GaussianSplatRenderer API and the
GaussianSplatAsset byte format.  See DESIGN.md / INTEGRATION.md."""
from .asset import (ColorFormat, GaussianSplatAsset, SHFormat, VectorFormat)  # noqa: F401

__version__ = "0.1.0"
