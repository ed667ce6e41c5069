// SPDX-License-Identifier: MIT
// GaussianSplatNative.cs -- the reference-side binding a maintainer adds to use libgsplat_hip.so from the C# host.
//
// This is SOURCE ONLY in this repository: the build image has no dotnet/mono, so it is not compiled here.  It mirrors
// include/gsplat_c.h 1:1 (the same table is bound and exercised through ctypes in unitygaussiansplatting_amd/_lib.py).
// Usage inside GaussianSplatRenderer.cs is shown in INTEGRATION.md: each block that records ComputeShader dispatches
// into a CommandBuffer (CreateResourcesForAsset :373-445, SortPoints :612-639, CalcViewData :579-610, the
// DrawProcedural of SortAndRenderSplats :156-166, the composite :206-209) becomes one call below.
using System;
using System.Runtime.InteropServices;

namespace GaussianSplatting.Runtime
{
    public static class GaussianSplatNative
    {
        const string Lib = "gsplat_hip";

        public enum Error { Ok = 0, InvalidArgument = -1, Hip = -2, UnsupportedFormat = -3, OutOfMemory = -4, InvalidAsset = -5, PairOverflow = -6, SortTimeout = -7, NoDevice = -8, Comm = -9 }
        public enum SortMode { Full = 0, Visible = 1 }       // gs_sort_mode: Full = SortPoints as the reference runs it; Visible = cull first, sort what is drawn

        [StructLayout(LayoutKind.Sequential)]
        public struct AssetDesc
        {
            public uint splatCount, posFormat, scaleFormat, colorFormat, shFormat, memoryKind;
            public IntPtr posData;   public ulong posSize;
            public IntPtr otherData; public ulong otherSize;
            public IntPtr colorData; public ulong colorSize;
            public IntPtr shData;    public ulong shSize;
            public IntPtr chunkData; public ulong chunkSize;
        }

        [StructLayout(LayoutKind.Sequential)]
        public unsafe struct FrameParams
        {
            public fixed float matrixMV[16];
            public fixed float matrixObjectToWorld[16];
            public fixed float matrixWorldToObject[16];
            public fixed float matrixVP[16];
            public float projM00, projM11;
            public float screenW, screenH;
            public fixed float camPosWorld[3];
            public float splatScale, opacityScale;
            public uint shOrder, shOnly;
            public float nearClip, farClip;
        }

        [StructLayout(LayoutKind.Sequential)]
        public struct FrameStats { public ulong tilePairs, pairCapacity; public uint visibleSplats, tilesX, tilesY, sortError, tileW, tileH, sortMode, tieLongRuns, tieLongestRun; }

        [StructLayout(LayoutKind.Sequential)]
        public struct StageTimes { public float calcDistancesMs, sortMs, calcViewMs, binMs, pairSortMs, blendMs, resolveMs, totalMs; public uint frames; public float onesweepDepthMs, onesweepPairsMs; public uint onesweepPairLaunches; public float onesweepDepthKernelMs, onesweepPairsKernelMs; public uint onesweepDepthLaunches; }

        [DllImport(Lib)] public static extern int gs_abi_version();
        [DllImport(Lib)] public static extern IntPtr gs_error_string(int err);
        [DllImport(Lib)] public static extern IntPtr gs_last_error_string();

        [DllImport(Lib)] public static extern int gs_context_create(int device, IntPtr hipStream, out IntPtr ctx);
        [DllImport(Lib)] public static extern int gs_context_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern int gs_context_synchronize(IntPtr ctx);
        [DllImport(Lib)] public static extern int gs_context_set_overlap(IntPtr ctx, int enabled);
        [DllImport(Lib)] public static extern int gs_context_set_shared_gpu(IntPtr ctx, int shared);
        [DllImport(Lib)] public static extern int gs_context_device_info(IntPtr ctx, byte[] nameOut, UIntPtr nameCap, out int cuCount, out ulong hbmBytes);

        [DllImport(Lib)] public static extern int gs_asset_create(IntPtr ctx, ref AssetDesc desc, out IntPtr asset);
        [DllImport(Lib)] public static extern int gs_asset_destroy(IntPtr asset);
        [DllImport(Lib)] public static extern int gs_asset_splat_count(IntPtr asset, out uint count);
        [DllImport(Lib)] public static extern int gs_asset_device_blobs(IntPtr asset, [Out] IntPtr[] ptrs5, [Out] ulong[] sizes5);
        [DllImport(Lib)] public static extern int gs_asset_info(IntPtr asset, [Out] uint[] out6);

        // multi-GPU (view-parallel): one context per GPU, the asset broadcast once over RCCL
        public const int GS_COMM_ID_BYTES = 128;
        [DllImport(Lib)] public static extern int gs_comm_unique_id([Out] byte[] id128);
        [DllImport(Lib)] public static extern int gs_comm_create(IntPtr ctx, int nranks, int rank, byte[] id128, out IntPtr comm);
        [DllImport(Lib)] public static extern int gs_comm_destroy(IntPtr comm);
        [DllImport(Lib)] public static extern int gs_comm_info(IntPtr comm, out int nranks, out int rank);
        [DllImport(Lib)] public static extern int gs_asset_broadcast(IntPtr comm, IntPtr assetOnRoot, int root, out IntPtr asset);
        [DllImport(Lib)] public static extern int gs_asset_replicate(IntPtr dstContext, IntPtr srcAsset, out IntPtr asset);

        [DllImport(Lib)] public static extern int gs_renderer_create(IntPtr ctx, IntPtr asset, out IntPtr renderer);
        [DllImport(Lib)] public static extern int gs_renderer_destroy(IntPtr renderer);
        [DllImport(Lib)] public static extern int gs_renderer_reset_order(IntPtr renderer);
        [DllImport(Lib)] public static extern int gs_renderer_sort(IntPtr renderer, float[] matrixSort16);
        [DllImport(Lib)] public static extern int gs_renderer_calc_view(IntPtr renderer, ref FrameParams p);
        [DllImport(Lib)] public static extern int gs_renderer_draw(IntPtr renderer, ref FrameParams p, IntPtr target);
        [DllImport(Lib)] public static extern int gs_renderer_render(IntPtr renderer, float[] matrixSort16, ref FrameParams p, IntPtr target, int doSort);
        // GaussianCutout.ShaderData (GaussianCutout.cs:19-23), matrix transposed to this ABI's row-major convention
        [StructLayout(LayoutKind.Sequential)]
        public unsafe struct Cutout { public fixed float matrix[16]; public uint typeAndFlags; }
        [DllImport(Lib)] public static extern int gs_renderer_set_cutouts(IntPtr renderer, Cutout[] cutouts, uint count);
        [DllImport(Lib)] public static extern int gs_renderer_set_deleted_bits(IntPtr renderer, uint[] words, UIntPtr wordCount);
        [DllImport(Lib)] public static extern int gs_renderer_set_view_buffer_mode(IntPtr renderer, int everyFrame);
        [DllImport(Lib)] public static extern int gs_renderer_set_blend_mode(IntPtr renderer, int mode);
        [DllImport(Lib)] public static extern int gs_renderer_set_tile_shape(IntPtr renderer, uint tileW, uint tileH);
        [DllImport(Lib)] public static extern int gs_renderer_tile_shape(IntPtr renderer, uint width, uint height, out uint tileW, out uint tileH);
        [DllImport(Lib)] public static extern int gs_renderer_set_profiling(IntPtr renderer, int frames);
        [DllImport(Lib)] public static extern int gs_renderer_set_kernel_timing(IntPtr renderer, int enabled);
        [DllImport(Lib)] public static extern int gs_renderer_reserve_pairs(IntPtr renderer, ulong pairCapacity);
        [DllImport(Lib)] public static extern int gs_renderer_poll_pairs(IntPtr renderer, out ulong tilePairs, out ulong pairCapacity);
        [DllImport(Lib)] public static extern int gs_renderer_download_order(IntPtr renderer, uint[] dst, UIntPtr count);
        [DllImport(Lib)] public static extern int gs_renderer_download_distances(IntPtr renderer, uint[] dst, UIntPtr count);
        [DllImport(Lib)] public static extern int gs_renderer_upload_order(IntPtr renderer, uint[] src, UIntPtr count);
        // gs_sort_mode: 0 = GS_SORT_FULL (SortPoints as the reference runs it), 1 = GS_SORT_VISIBLE (cull first, sort what is drawn)
        [DllImport(Lib)] public static extern int gs_renderer_set_sort_mode(IntPtr renderer, int mode);
        [DllImport(Lib)] public static extern int gs_renderer_sort_mode(IntPtr renderer, out int mode, out int active);
        [DllImport(Lib)] public static extern int gs_renderer_set_sort_history_limit(IntPtr renderer, uint rows);
        [DllImport(Lib)] public static extern int gs_renderer_sort_history(IntPtr renderer, out uint rows, out uint limit, out ulong consolidations);
        [DllImport(Lib)] public static extern int gs_renderer_set_frames_in_flight(IntPtr renderer, int frames);
        [DllImport(Lib)] public static extern int gs_renderer_frames_in_flight(IntPtr renderer, out int frames, out int active);
        [DllImport(Lib)] public static extern int gs_renderer_download_visible_order(IntPtr renderer, uint[] dst, UIntPtr capacity, out uint count);
        [DllImport(Lib)] public static extern int gs_renderer_set_render_mode(IntPtr renderer, int mode, float pointDisplaySize);
        [DllImport(Lib)] public static extern int gs_renderer_download_view(IntPtr renderer, IntPtr dst, UIntPtr bytes);
        [DllImport(Lib)] public static extern int gs_renderer_download_raster_records(IntPtr renderer, IntPtr recs, IntPtr rects, IntPtr visMask);
        [DllImport(Lib)] public static extern int gs_renderer_frame_stats(IntPtr renderer, out FrameStats stats);
        [DllImport(Lib)] public static extern int gs_renderer_frame_times(IntPtr renderer, [Out] float[] ms, int capacity, out int count);
        [DllImport(Lib)] public static extern int gs_renderer_stage_times(IntPtr renderer, out StageTimes times);

        [DllImport(Lib)] public static extern int gs_target_create(IntPtr ctx, uint width, uint height, out IntPtr target);
        [DllImport(Lib)] public static extern int gs_target_destroy(IntPtr target);
        [DllImport(Lib)] public static extern int gs_target_clear(IntPtr target);
        [DllImport(Lib)] public static extern int gs_target_download(IntPtr target, IntPtr dstRgba16f, UIntPtr bytes);
        [DllImport(Lib)] public static extern int gs_target_resolve(IntPtr target, float[] backgroundRgba4, IntPtr dstRgba32f, IntPtr dstRgba8);
        [DllImport(Lib)] public static extern int gs_target_set_scene_depth(IntPtr target, IntPtr depth, int memoryKind);
        [DllImport(Lib)] public static extern int gs_target_device_ptr(IntPtr target, out IntPtr rgba16fDev, out IntPtr resolvedDev);
        [DllImport(Lib)] public static extern int gs_target_set_profiling(IntPtr target, int enabled);
        [DllImport(Lib)] public static extern int gs_target_resolve_time(IntPtr target, out float meanMs, out int count);

        // native importer (host code): GaussianSplatAssetCreator.CreateAsset without the asset database
        [StructLayout(LayoutKind.Sequential)]
        public struct ImportInput { public uint splatCount; public IntPtr pos, dc0, sh, opacity, scale, rot; }
        [StructLayout(LayoutKind.Sequential)]
        public struct ImportFormats { public uint posFormat, scaleFormat, colorFormat, shFormat, linearize, morton; }
        [DllImport(Lib)] public static extern int gs_import_blob_sizes(uint splatCount, ref ImportFormats formats, [Out] ulong[] sizes5);
        [DllImport(Lib)] public static extern int gs_import_encode(ref ImportInput input, ref ImportFormats formats, IntPtr[] blobs5, ulong[] sizes5, float[] boundsMin3, float[] boundsMax3);

        [DllImport(Lib)] public static extern int gs_ply_open([MarshalAs(UnmanagedType.LPStr)] string path, out IntPtr ply, out uint splatCount);
        [DllImport(Lib)] public static extern int gs_spz_open([MarshalAs(UnmanagedType.LPStr)] string path, out IntPtr ply, out uint splatCount);
        [DllImport(Lib)] public static extern int gs_ply_arrays(IntPtr ply, out ImportInput arrays);
        [DllImport(Lib)] public static extern int gs_ply_close(IntPtr ply);

        [DllImport(Lib)] public static extern int gs_sorter_create(IntPtr ctx, uint maxCount, out IntPtr sorter);
        [DllImport(Lib)] public static extern int gs_sorter_destroy(IntPtr sorter);
        [DllImport(Lib)] public static extern int gs_sorter_dispatch(IntPtr sorter, IntPtr keysDev, IntPtr valuesDev, uint count, uint keyBits);
        [DllImport(Lib)] public static extern int gs_sorter_sort_host(IntPtr sorter, uint[] keys, uint[] values, uint count, uint keyBits);

        public static void Check(int rc, string where)
        {
            if (rc != 0)
                throw new InvalidOperationException($"{where}: {(Error)rc} ({Marshal.PtrToStringAnsi(gs_last_error_string())})");
        }
    }
}
