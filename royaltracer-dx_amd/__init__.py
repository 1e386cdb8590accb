"""royaltracer-dx_amd — ctypes binding of the MI355X wavefront path tracer (librtx_hip.so).

The product path is the HIP library only.  Importing this package raises ImportError if
librtx_hip.so has not been built, and Context() raises RtxError if no GPU can be opened:
there is no CPU fallback and nothing under oracle/ is ever imported from here.

The C-ABI is declared in include/rtx.h (hot path) and include/rtx_host.h (scene-loader /
material / camera layer mirroring the reference's ObjLoader / Manipulator / Renderer).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_LIB_PATH") or os.path.join(_HERE, "librtx_hip.so")   # RTX_LIB_PATH: tooling builds (make PROFILE=1)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (hipcc --offload-arch=gfx950) "
        "or `python -c 'import __graft_entry__ as g; g.build()'`.  There is no CPU fallback.")

# PyTorch wheels bundle their own libamdhip64.so.7.  A process must hold ONE HIP runtime, otherwise the second
# one to initialise sees no devices and stream / device-pointer interop (rtx_set_stream, rtx_bind_accum) is
# meaningless.  Import torch first (when present) so that librtx_hip.so binds to the runtime torch uses.
if os.environ.get("RTX_NO_TORCH_PRELOAD", "0") != "1":
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional: the library only needs the HIP runtime
        pass

lib = C.CDLL(LIB_PATH)

RTX_OK = 0
FLAG_LAMBERT_ONLY = 1
FLAG_JITTER = 2
FLAG_TRANSMISSION = 4
FLAG_BLOCK_TILES = 8
K_RAYGEN, K_TRACE, K_SHADE, K_SHADOW, K_ACCUM, K_SORT, K_BOUNCE, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 8
KERNEL_NAMES = {K_RAYGEN: "raygen", K_TRACE: "trace_closest", K_SHADE: "shade", K_SHADOW: "trace_shadow", K_ACCUM: "accumulate", K_SORT: "sort", K_BOUNCE: "bounce_fused"}
OPT_KERNEL_TIMING, OPT_PATHS_PER_BATCH, OPT_SORT_MATERIALS, OPT_LDS_NODES, OPT_SMALL_SCENE, OPT_FUSED_BOUNCE, OPT_BOUNCE_VARIANT = 1, 2, 3, 4, 5, 6, 7
OPT_REFILL_MIN, OPT_STACK_PRIVATE, OPT_TRACE_SCHED, OPT_GPU_REFIT, OPT_BLOCKS_PER_CU, OPT_LPT_ORDER, OPT_FUSED_BVH, OPT_WORK_STEALING, OPT_COMPACT_STATE, OPT_OVERLAP_SHADOW = 8, 9, 10, 11, 12, 13, 14, 15, 16, 18
OPT_RESTIR_WAVEFRONT, OPT_RESTIR_CHUNKS, OPT_OCCLUDER_CACHE, OPT_RESTIR_LANES, OPT_SHADE_DENSE, OPT_MERGE_RAYS, OPT_TAPER = 19, 20, 21, 22, 23, 24, 25
OPT_BVH_REINSERT, OPT_BVH_SPLIT, OPT_ANYHIT_ORDER, OPT_RESTIR_LANE_MIN, OPT_TRACE_COUNTERS, OPT_ASYNC, OPT_OCTANT_SORT, OPT_SAMPLE_INTERLEAVE, OPT_NODE_STRIDE, OPT_RESTIR_KEYS, OPT_LDS_NODES_CLOSEST, OPT_PARTIAL_REFIT = 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37
OPT_GPU_BUILD = 38
OPT_STACK_CAP = 39


class RtxError(RuntimeError):
    pass


class Params(C.Structure):
    """rtx_params (include/rtx.h)."""
    _fields_ = [(n, C.c_uint32) for n in (
        "width", "height", "spp", "sample_base", "max_bounces", "nee_samples", "rr_start",
        "frame_seed", "flags", "tile_size", "shard_rank", "shard_count")]

    def __init__(self, width=1920, height=1080, spp=1, sample_base=1, max_bounces=8, nee_samples=1,
                 rr_start=3, frame_seed=1, flags=0, tile_size=64, shard_rank=0, shard_count=1):
        super().__init__(width, height, spp, sample_base, max_bounces, nee_samples, rr_start,
                         frame_seed, flags, tile_size, shard_rank, shard_count)

    def copy(self, **kw):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d.update(kw)
        return Params(**d)


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_extension", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("paths", C.c_uint64), ("kernel_ms", C.c_double * K_COUNT), ("kernel_launches", C.c_uint64 * K_COUNT),
                ("kernel_items", C.c_uint64 * K_COUNT), ("render_ms", C.c_double),
                ("bvh_nodes", C.c_uint32), ("triangles", C.c_uint32), ("lights", C.c_uint32), ("materials", C.c_uint32),
                ("primary_hits", C.c_uint64), ("bvh_refits", C.c_uint32), ("bvh_refs", C.c_uint32), ("restir_stale_history_reads", C.c_uint64)]

    @property
    def rays(self):
        return self.rays_primary + self.rays_extension + self.rays_shadow


_vp, _u32, _fp = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


# ---- include/rtx.h ----
_sig("rtx_create", C.c_int, C.c_int, C.POINTER(_vp))
_sig("rtx_destroy", None, _vp)
_sig("rtx_last_error", C.c_char_p, _vp)
_sig("rtx_set_option", C.c_int, _vp, C.c_int, C.c_int64)
_sig("rtx_set_stream", C.c_int, _vp, _vp)
_sig("rtx_set_materials", C.c_int, _vp, _vp, _u32)
_sig("rtx_add_mesh", C.c_int, _vp, _vp, _u32, _vp, _u32, _vp, _u32p)
_sig("rtx_add_instance", C.c_int, _vp, _u32, _fp, _u32p)
_sig("rtx_set_instance_transform", C.c_int, _vp, _u32, _fp)
_sig("rtx_commit_scene", C.c_int, _vp)
_sig("rtx_set_camera", C.c_int, _vp, _fp, _fp)
_sig("rtx_save_scene_cache", C.c_int, _vp, C.c_char_p)
_sig("rtx_load_scene_cache", C.c_int, _vp, C.c_char_p)
_sig("rtx_bind_accum", C.c_int, _vp, _vp, C.c_size_t)
_sig("rtx_clear_accum", C.c_int, _vp, _u32, _u32)
_sig("rtx_render", C.c_int, _vp, C.POINTER(Params))
_sig("rtx_read_accum", C.c_int, _vp, _vp, C.c_size_t)
_sig("rtx_render_v6_pass1", C.c_int, _vp, C.POINTER(Params))
_sig("rtx_render_restir", C.c_int, _vp, C.POINTER(Params))
_sig("rtx_restir_reset", C.c_int, _vp)
_sig("rtx_restir_state_slab_bytes", C.c_int, C.POINTER(Params), C.POINTER(C.c_size_t))
_sig("rtx_restir_pack_state", C.c_int, _vp, C.POINTER(Params), _vp)
_sig("rtx_restir_unpack_state", C.c_int, _vp, C.POINTER(Params), _vp)
class HaloPeer(C.Structure):
    """rtx_halo_peer (include/rtx.h): one neighbour of a rank in the halo exchange of the ReSTIR history"""
    _fields_ = [(n, C.c_uint32) for n in ("rank", "send_x0", "send_y0", "send_x1", "send_y1", "recv_x0", "recv_y0", "recv_x1", "recv_y1")] + \
               [(n, C.c_uint64) for n in ("send_offset", "send_bytes", "recv_offset", "recv_bytes")]


_sig("rtx_restir_halo_plan", C.c_int, C.POINTER(Params), _u32, C.POINTER(HaloPeer), _u32, _u32p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_sig("rtx_restir_pack_halo", C.c_int, _vp, C.POINTER(Params), _u32, _vp)
_sig("rtx_restir_unpack_halo", C.c_int, _vp, C.POINTER(Params), _u32, _vp)
_sig("rtx_debug_tree_hash", C.c_int, _vp, C.POINTER(C.c_uint64))
_sig("rtx_debug_read_host_build", C.c_int, _vp, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
_sig("rtx_debug_host_checksums", C.c_int, _vp, C.POINTER(C.c_uint64))
_sig("rtx_debug_read_tree", C.c_int, _vp, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64)
_sig("rtx_debug_build_info", C.c_int, _vp, C.POINTER(C.c_double), _u32p)
_sig("rtx_read_restir_last", C.c_int, _vp, _vp, _vp, _vp, C.c_size_t)
_sig("rtx_pass1_slots", C.c_size_t, _u32, _u32)
_sig("rtx_read_pass1_buffers", C.c_int, _vp, _vp, _vp, _vp, C.c_size_t)
_sig("rtx_read_srgb8", C.c_int, _vp, _vp, C.c_size_t)
_sig("rtx_get_stats", C.c_int, _vp, C.POINTER(Stats))
_sig("rtx_read_layer", C.c_int, _vp, _u32, _u32, _u32, _vp, C.c_size_t)
_sig("rtx_get_lights", C.c_int, _vp, _vp, _u32, _u32p)
_sig("rtx_shard_slab_bytes", C.c_int, C.POINTER(Params), C.POINTER(C.c_size_t))
_sig("rtx_pack_tiles", C.c_int, _vp, C.POINTER(Params), _vp)
_sig("rtx_unpack_tiles", C.c_int, _vp, C.POINTER(Params), _vp)
_sig("rtx_debug_primary_rays", C.c_int, _vp, C.POINTER(Params), _u32, _vp)
_sig("rtx_debug_trace_closest", C.c_int, _vp, _vp, _u32, _vp)
_sig("rtx_debug_trace_any", C.c_int, _vp, _vp, _u32, _vp)
_sig("rtx_debug_trace_stats", C.c_int, _vp, _vp, _u32, _vp)
_sig("rtx_debug_validate_bvh", C.c_int, _vp)
_sig("rtx_debug_trace_counters", C.c_int, _vp, C.POINTER(C.c_uint64))
_sig("rtx_debug_surface", C.c_int, _vp, _vp, _vp, _u32, _vp)
_sig("rtx_debug_bsdf_eval", C.c_int, _vp, _u32, _u32, _vp, _u32, _vp)
_sig("rtx_debug_bsdf_sample", C.c_int, _vp, _u32, _u32, _vp, _u32, _vp)
_sig("rtx_debug_tea", C.c_int, _vp, _u32p, _u32, _vp)
# ---- include/rtx_host.h ----
_sig("rtxh_scene_cornell", _vp)
_sig("rtxh_scene_sponza_class", _vp, _u32, _u32)
_sig("rtxh_scene_bistro_class", _vp, _u32, _u32)
_sig("rtxh_scene_sponza_class_hard", _vp, _u32, _u32)
_sig("rtxh_scene_bistro_class_hard", _vp, _u32, _u32)
_sig("rtxh_scene_from_obj", _vp, C.POINTER(C.c_char_p), _u32, C.c_char_p)
_sig("rtxh_scene_free", None, _vp)
_sig("rtxh_scene_save", C.c_int, _vp, C.c_char_p)
_sig("rtxh_scene_load", _vp, C.c_char_p)
_sig("rtxh_last_error", C.c_char_p)
_sig("rtxh_scene_num_materials", _u32, _vp)
_sig("rtxh_scene_materials", _vp, _vp)
class MaterialExt(C.Structure):
    """rtxh_material_ext (include/rtx_host.h)"""
    _fields_ = [("Ni", C.c_float), ("Ns", C.c_float), ("Pcr", C.c_float), ("aniso", C.c_float), ("anisor", C.c_float), ("illum", C.c_int32),
                ("Ka", C.c_float * 3), ("Tf", C.c_float * 3), ("map", C.c_int32 * 13)]


MAP_SLOTS = ("Ka", "Kd", "Ks", "Ke", "Ns", "bump", "d", "disp", "refl", "Pr", "Pm", "Ps", "norm")
_sig("rtxh_scene_material_ext", C.c_int, _vp, _u32, C.POINTER(MaterialExt))
_sig("rtxh_scene_num_textures", _u32, _vp)
_sig("rtxh_scene_texture", C.c_char_p, _vp, _u32)
_sig("rtxh_scene_num_meshes", _u32, _vp)
_sig("rtxh_scene_mesh", C.c_int, _vp, _u32, C.POINTER(_vp), _u32p, C.POINTER(_vp), _u32p, C.POINTER(_vp))
_sig("rtxh_scene_num_instances", _u32, _vp)
_sig("rtxh_scene_instance", C.c_int, _vp, _u32, _u32p, _fp)
_sig("rtxh_scene_num_triangles", C.c_uint64, _vp)
_sig("rtxh_scene_camera", C.c_int, _vp, _fp, _fp, _fp, _fp, _fp, _fp)
_sig("rtxh_scene_view_proj", C.c_int, _vp, C.c_float, _fp, _fp)
_sig("rtxh_scene_upload", C.c_int, _vp, _vp, C.c_float)
_sig("rtxh_lookat", None, _fp, _fp, _fp, _fp)
_sig("rtxh_perspective_fov_rh", None, C.c_float, C.c_float, C.c_float, C.c_float, _fp)
_sig("rtxh_generate_ess_lut", None, C.c_float, _fp)
_sig("rtxh_mat4_inverse", None, _fp, _fp)
_sig("rtxh_half_round", C.c_float, C.c_float)
_sig("rtxh_bvh_check", C.c_int, _vp, _u32, _u32p, _u32p, _u32p)
_sig("rtxh_bvh_refit_check", C.c_int, _vp, _vp, _u32)
_sig("rtxh_bvh8_check", C.c_int, _vp, _u32, _u32p, _u32p)
_sig("rtxh_bvh8_stats", C.c_int, _vp, _u32, _vp, _u32p)
_sig("rtxh_bvh_option", C.c_int, C.c_char_p, C.c_double)
_sig("rtxh_bvh_replay", C.c_int, _vp, _u32, _vp, _u32, C.c_int, _u32, _vp, _u32p)
_sig("rtxh_scene_small_occluders", C.c_int, _vp, _u32p)
_sig("rtxh_scene_anyhit_order", C.c_int, _vp, _u32p)
_sig("rtxh_write_png", C.c_int, C.c_char_p, _vp, _u32, _u32)
_sig("rtxh_write_ppm", C.c_int, C.c_char_p, _vp, _u32, _u32)
_sig("rtxh_write_exr", C.c_int, C.c_char_p, _vp, _u32, _u32)
_sig("rtxh_scene_small_records", C.c_int, _vp, _vp, _vp, _u32, _u32p, _fp, _fp)
_sig("rtxh_scene_set_camera", C.c_int, _vp, _fp, _fp, _fp)
_sig("rtxh_renderer_create", _vp, _u32, _u32, C.c_char_p, C.c_int)
_sig("rtxh_renderer_set_scene", C.c_int, _vp, _vp)
_sig("rtxh_renderer_params", C.POINTER(Params), _vp)
_sig("rtxh_renderer_restir_params", C.POINTER(Params), _vp)
_sig("rtxh_renderer_set_mode", C.c_int, _vp, C.c_int)
_sig("rtxh_renderer_context", _vp, _vp)
_sig("rtxh_renderer_on_init", C.c_int, _vp)
_sig("rtxh_renderer_on_update", C.c_int, _vp)
_sig("rtxh_renderer_set_instance_transform", C.c_int, _vp, _u32, _vp)
_sig("rtxh_renderer_on_render", C.c_int, _vp)
_sig("rtxh_renderer_read_accum", C.c_int, _vp, _vp, C.c_size_t)
_sig("rtxh_renderer_read_output", C.c_int, _vp, _vp, C.c_size_t)
_sig("rtxh_renderer_on_key_up", C.c_int, _vp, C.c_uint8)
_sig("rtxh_renderer_display_layer", _u32, _vp)
_sig("rtxh_renderer_destroy", None, _vp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(_vp)


def _fptr(a):
    return a.ctypes.data_as(_fp)


# ------------------------------------------------------------------------------------------------
# host layer
# ------------------------------------------------------------------------------------------------
class Scene:
    """A host-side scene in the reference's data model (global Material table, per-model
    Vertex / index / materialID arrays, instance list).  Arrays are numpy copies."""

    def __init__(self, handle):
        if not handle:
            raise RtxError("scene creation failed: " + lib.rtxh_last_error().decode())
        self._h = handle
        n = lib.rtxh_scene_num_materials(handle)
        buf = (C.c_float * (n * 32)).from_address(lib.rtxh_scene_materials(handle)) if n else []
        self.materials = np.array(buf, dtype=np.float32).reshape(n, 32)
        self.textures = [lib.rtxh_scene_texture(handle, i).decode() for i in range(lib.rtxh_scene_num_textures(handle))]
        self.material_ext = []                       # per material (OBJ / MTL scenes only): the MTL fields and map ids beside the 128-byte record
        for i in range(n):
            x = MaterialExt()
            if lib.rtxh_scene_material_ext(handle, i, C.byref(x)) != RTX_OK:
                break
            self.material_ext.append(dict(Ni=x.Ni, Ns=x.Ns, Pcr=x.Pcr, aniso=x.aniso, anisor=x.anisor, illum=x.illum, Ka=list(x.Ka), Tf=list(x.Tf),
                                          maps={MAP_SLOTS[k]: self.textures[x.map[k]] for k in range(13) if x.map[k] >= 0}))
        self.meshes = []
        for i in range(lib.rtxh_scene_num_meshes(handle)):
            v, idx, mid = _vp(), _vp(), _vp()
            nv, ni = _u32(), _u32()
            lib.rtxh_scene_mesh(handle, i, C.byref(v), C.byref(nv), C.byref(idx), C.byref(ni), C.byref(mid))
            verts = np.array((C.c_float * (nv.value * 7)).from_address(v.value), dtype=np.float32).reshape(-1, 7) if nv.value else np.zeros((0, 7), np.float32)
            indices = np.array((C.c_uint32 * ni.value).from_address(idx.value), dtype=np.uint32) if ni.value else np.zeros(0, np.uint32)
            matids = np.array((C.c_uint32 * ni.value).from_address(mid.value), dtype=np.uint32) if ni.value else np.zeros(0, np.uint32)
            self.meshes.append((verts, indices, matids))
        self.instances = []
        for i in range(lib.rtxh_scene_num_instances(handle)):
            mesh = _u32()
            m = np.zeros(16, np.float32)
            lib.rtxh_scene_instance(handle, i, C.byref(mesh), _fptr(m))
            self.instances.append((mesh.value, m))
        self.num_triangles = lib.rtxh_scene_num_triangles(handle)
        e, c, u = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        fov, zn, zf = C.c_float(), C.c_float(), C.c_float()
        lib.rtxh_scene_camera(handle, _fptr(e), _fptr(c), _fptr(u), C.byref(fov), C.byref(zn), C.byref(zf))
        self.eye, self.center, self.up, self.fovy_deg, self.znear, self.zfar = e, c, u, fov.value, zn.value, zf.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:                      # module globals are gone at interpreter shutdown
            lib.rtxh_scene_free(h)

    @classmethod
    def cornell(cls):
        return cls(lib.rtxh_scene_cornell())

    @classmethod
    def sponza_class(cls, target_tris=262144, seed=260, hard=False):
        """the Sponza-class atrium; hard=True: the size distribution of the real asset (host/Scenes.h)"""
        return cls((lib.rtxh_scene_sponza_class_hard if hard else lib.rtxh_scene_sponza_class)(target_tris, seed))

    @classmethod
    def bistro_class(cls, target_tris=3800000, seed=3800, hard=False):
        return cls((lib.rtxh_scene_bistro_class_hard if hard else lib.rtxh_scene_bistro_class)(target_tris, seed))

    @classmethod
    def from_obj(cls, files, mtl_dir):
        arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
        return cls(lib.rtxh_scene_from_obj(arr, len(files), mtl_dir.encode()))

    def set_camera(self, eye, center, up=(0.0, 1.0, 0.0)):
        e, c, u = (np.asarray(v, np.float32) for v in (eye, center, up))
        lib.rtxh_scene_set_camera(self._h, _fptr(e), _fptr(c), _fptr(u))
        self.eye, self.center, self.up = e, c, u

    def bounds(self):
        """world-space bounding box of all instances -> (lo, hi)"""
        lo, hi = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        for mesh, m in self.instances:
            v = self.meshes[mesh][0][:, :3]
            if len(v):
                M = np.asarray(m, np.float32).reshape(4, 4)              # column-major 16 floats: element (r, c) at m[c * 4 + r]
                w = v @ M[:3, :3] + M[3, :3]
                lo, hi = np.minimum(lo, w.min(0)), np.maximum(hi, w.max(0))
        return lo, hi

    def frame_bounds(self):
        """camera for a model that brings none (an OBJ asset): inside the bounding box at 40 % of its height, 35 % of the way in from one end of its longer horizontal
        axis, looking along that axis (Sponza: down the nave)"""
        lo, hi = self.bounds()
        c, ext = (lo + hi) * 0.5, hi - lo
        ax = 0 if ext[0] >= ext[2] else 2
        eye, ctr = c.copy(), c.copy()
        eye[1] = ctr[1] = lo[1] + 0.4 * ext[1]
        eye[ax] = c[ax] - 0.35 * ext[ax]
        self.set_camera(eye, ctr)

    def save(self, path):
        """write the binary scene cache: this scene + its BVH / shading records / LUTs (built on the host, no GPU needed)"""
        if lib.rtxh_scene_save(self._h, str(path).encode()) != RTX_OK:
            raise RtxError("rtxh_scene_save failed: " + lib.rtxh_last_error().decode())

    @classmethod
    def load(cls, path):
        """a scene from a binary cache file; Context.upload(scene) then hands the prebuilt arrays to the GPU without rebuilding"""
        s = cls(lib.rtxh_scene_load(str(path).encode()))
        s.cache_path = str(path)
        return s

    def small_records(self):
        """-> (records (n,20) f32, triangle ids (n,2) i32, delta, cm) of the tiny-scene pre-test, or None"""
        recs, ids = np.zeros((64, 20), np.float32), np.zeros((64, 2), np.int32)
        n, d, cm = _u32(), C.c_float(), C.c_float()
        if lib.rtxh_scene_small_records(self._h, _ptr(recs), _ptr(ids), 64, C.byref(n), C.byref(d), C.byref(cm)) != RTX_OK or n.value == 0:
            return None
        return recs[:n.value], ids[:n.value], d.value, cm.value

    def small_occluders(self):
        """number of leading tiny-scene records that are NOT faces of the scene's convex hull"""
        n = _u32()
        if lib.rtxh_scene_small_occluders(self._h, C.byref(n)) != RTX_OK:
            raise RtxError("rtxh_scene_small_occluders failed")
        return n.value

    def anyhit_order(self):
        """the any-hit visiting order the commit-time probe picks for this scene (0 slot order, 1 nearest octant first, 2 farthest first); host only"""
        n = _u32()
        if lib.rtxh_scene_anyhit_order(self._h, C.byref(n)) != RTX_OK:
            raise RtxError("rtxh_scene_anyhit_order failed")
        return n.value

    def view_proj(self, aspect):
        v, p = np.zeros(16, np.float32), np.zeros(16, np.float32)
        lib.rtxh_scene_view_proj(self._h, C.c_float(aspect), _fptr(v), _fptr(p))
        return v, p


def lookat(eye, center, up):
    e, c, u, v = _f32(eye), _f32(center), _f32(up), np.zeros(16, np.float32)
    lib.rtxh_lookat(_fptr(e), _fptr(c), _fptr(u), _fptr(v))
    return v


def perspective_fov_rh(fovy_rad, aspect, zn, zf):
    p = np.zeros(16, np.float32)
    lib.rtxh_perspective_fov_rh(fovy_rad, aspect, zn, zf, _fptr(p))
    return p


def generate_ess_lut(roughness):
    lut = np.zeros(16, np.float32)
    lib.rtxh_generate_ess_lut(roughness, _fptr(lut))
    return lut


def mat4_inverse(m):
    m, o = _f32(m).reshape(16), np.zeros(16, np.float32)
    lib.rtxh_mat4_inverse(_fptr(m), _fptr(o))
    return o


def half_round(x):
    return lib.rtxh_half_round(C.c_float(x))


def bvh_refit_check(before, after):
    a, b = _f32(before).reshape(-1, 9), _f32(after).reshape(-1, 9)
    return lib.rtxh_bvh_refit_check(_ptr(a), _ptr(b), len(a))


def bvh_check(world_tris):
    w = _f32(world_tris).reshape(-1, 9)
    nodes, depth, leaf = _u32(), _u32(), _u32()
    rc = lib.rtxh_bvh_check(_ptr(w), len(w), C.byref(nodes), C.byref(depth), C.byref(leaf))
    return rc, nodes.value, depth.value, leaf.value


def write_image(path, image):
    """(H, W, 4) uint8 sRGB -> .png / .ppm; (H, W, 4) float32 accumulation buffer -> .exr (xyz / count)"""
    a = np.ascontiguousarray(image)
    h, w = a.shape[:2]
    if path.endswith(".exr"):
        rc = lib.rtxh_write_exr(path.encode(), _ptr(_f32(a)), w, h)
    else:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        rc = (lib.rtxh_write_ppm if path.endswith(".ppm") else lib.rtxh_write_png)(path.encode(), _ptr(a), w, h)
    if rc != 0:
        raise RtxError("could not write " + path)


def bvh8_stats(world_tris):
    """(hist of leaf-slot sizes 0..4 + internal slots, wide node count) of the device tree built for these triangles"""
    w = _f32(world_tris).reshape(-1, 9)
    hist = np.zeros(6, np.uint32); nodes = _u32()
    rc = lib.rtxh_bvh8_stats(_ptr(w), len(w), _ptr(hist), C.byref(nodes))
    return rc, hist, nodes.value


def restir_halo_plan(params, halo_px):
    """rtx_restir_halo_plan (no context, no GPU): ([HaloPeer ...] in ascending rank order, send bytes, receive bytes) of params.shard_rank in the RTX_FLAG_BLOCK_TILES deal"""
    peers = (HaloPeer * 64)(); n = _u32(); st, rt_ = C.c_uint64(), C.c_uint64()
    if lib.rtx_restir_halo_plan(C.byref(params), halo_px, peers, 64, C.byref(n), C.byref(st), C.byref(rt_)) != 0:
        raise RtxError("rtx_restir_halo_plan: " + (lib.rtx_last_error(None) or b"").decode())
    return [peers[i] for i in range(n.value)], st.value, rt_.value


def bvh_option(key, value):
    """process-wide builder default (csrc/rtx_scene_host.hpp BvhBuildOptions); contexts created afterwards start with it"""
    if lib.rtxh_bvh_option(key.encode(), float(value)) != 0:
        raise RtxError("unknown BVH builder option " + key)


def bvh_replay(world_tris, rays8, any_hit=False, any_order=0):
    """host replay of the device traversal on the tree the current builder options give: (hits (n, 4): t, node steps, triangle tests, id bits; leaf entries)"""
    w = _f32(world_tris).reshape(-1, 9); r = _f32(rays8).reshape(-1, 8)
    out = np.zeros((len(r), 4), np.float32); refs = _u32()
    rc = lib.rtxh_bvh_replay(_ptr(w), len(w), _ptr(r), len(r), int(any_hit), any_order, _ptr(out), C.byref(refs))
    if rc != 0:
        raise RtxError("rtxh_bvh_replay failed: %d" % rc)
    return out, refs.value


def bvh8_check(world_tris):
    w = _f32(world_tris).reshape(-1, 9)
    nodes, stack = _u32(), _u32()
    rc = lib.rtxh_bvh8_check(_ptr(w), len(w), C.byref(nodes), C.byref(stack))
    return rc, nodes.value, stack.value


# ------------------------------------------------------------------------------------------------
# the hot path
# ------------------------------------------------------------------------------------------------
class Context:
    """One rtx_ctx (one GPU)."""

    def __init__(self, device=0):
        h = _vp()
        rc = lib.rtx_create(device, C.byref(h))
        if rc != RTX_OK:
            raise RtxError(f"rtx_create({device}) failed ({rc}): {lib.rtx_last_error(None).decode()}")
        self._h = h
        self.device = device
        self.width = self.height = 0

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:          # (at interpreter shutdown the module globals may already be gone: the process's exit frees the context)
            lib.rtx_destroy(h)

    __del__ = close

    def _ck(self, rc, what):
        if rc != RTX_OK:
            raise RtxError(f"{what} failed ({rc}): {lib.rtx_last_error(self._h).decode()}")

    def set_option(self, opt, value):
        self._ck(lib.rtx_set_option(self._h, opt, int(value)), "rtx_set_option")

    def set_stream(self, stream_handle):
        self._ck(lib.rtx_set_stream(self._h, _vp(stream_handle) if stream_handle else None), "rtx_set_stream")

    def set_materials(self, mats):
        m = _f32(mats).reshape(-1, 32)
        self._ck(lib.rtx_set_materials(self._h, _ptr(m), len(m)), "rtx_set_materials")

    def add_mesh(self, verts, indices, matids):
        v = _f32(verts).reshape(-1, 7)
        i = np.ascontiguousarray(indices, dtype=np.uint32)
        m = np.ascontiguousarray(matids, dtype=np.uint32)
        out = _u32()
        self._ck(lib.rtx_add_mesh(self._h, _ptr(v), len(v), _ptr(i), len(i), _ptr(m), C.byref(out)), "rtx_add_mesh")
        return out.value

    def add_instance(self, mesh, o2w):
        m = _f32(o2w).reshape(16)
        out = _u32()
        self._ck(lib.rtx_add_instance(self._h, mesh, _fptr(m), C.byref(out)), "rtx_add_instance")
        return out.value

    def set_instance_transform(self, inst, o2w):
        m = _f32(o2w).reshape(16)
        self._ck(lib.rtx_set_instance_transform(self._h, inst, _fptr(m)), "rtx_set_instance_transform")

    def commit(self):
        self._ck(lib.rtx_commit_scene(self._h), "rtx_commit_scene")

    def set_camera(self, view, proj):
        v, p = _f32(view).reshape(16), _f32(proj).reshape(16)
        self._ck(lib.rtx_set_camera(self._h, _fptr(v), _fptr(p)), "rtx_set_camera")

    def save_scene_cache(self, path):
        self._ck(lib.rtx_save_scene_cache(self._h, str(path).encode()), "rtx_save_scene_cache")

    def load_scene_cache(self, path):
        self._ck(lib.rtx_load_scene_cache(self._h, str(path).encode()), "rtx_load_scene_cache")

    def upload(self, scene, aspect):
        """rtx_set_materials / add_mesh / add_instance / commit / set_camera from a Scene (array path); a Scene.load()-ed scene
        goes through rtx_load_scene_cache instead (prebuilt BVH)."""
        if getattr(scene, "cache_path", None):
            self.load_scene_cache(scene.cache_path)
            self.set_camera(*scene.view_proj(aspect))
            return
        self.set_materials(scene.materials)
        for v, i, m in scene.meshes:
            self.add_mesh(v, i, m)
        for mesh, o2w in scene.instances:
            self.add_instance(mesh, o2w)
        self.commit()
        self.set_camera(*scene.view_proj(aspect))

    def bind_accum(self, device_ptr, nbytes):
        self._ck(lib.rtx_bind_accum(self._h, _vp(device_ptr) if device_ptr else None, nbytes), "rtx_bind_accum")

    def clear(self, width, height):
        self.width, self.height = width, height
        self._ck(lib.rtx_clear_accum(self._h, width, height), "rtx_clear_accum")

    def render(self, params):
        self.width, self.height = params.width, params.height
        self._ck(lib.rtx_render(self._h, C.byref(params)), "rtx_render")

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._ck(lib.rtx_read_accum(self._h, _ptr(out), out.nbytes), "rtx_read_accum")
        return out

    def render_v6_pass1(self, params):
        """the reference's own pass 1 (RIS direct light + SamplePathSimple); see rtx_render_v6_pass1"""
        self.width, self.height = params.width, params.height
        self._ck(lib.rtx_render_v6_pass1(self._h, C.byref(params)), "rtx_render_v6_pass1")

    def render_restir(self, params):
        """`params.spp` consecutive ReSTIR frames (pass 1 + temporal + spatial) with the current camera"""
        self.width, self.height = params.width, params.height
        self._ck(lib.rtx_render_restir(self._h, C.byref(params)), "rtx_render_restir")

    def restir_state_slab_bytes(self, params):
        b = C.c_size_t()
        self._ck(lib.rtx_restir_state_slab_bytes(C.byref(params), C.byref(b)), "rtx_restir_state_slab_bytes")
        return b.value

    def restir_pack_state(self, params, device_ptr):
        self._ck(lib.rtx_restir_pack_state(self._h, C.byref(params), _vp(device_ptr)), "rtx_restir_pack_state")

    def restir_unpack_state(self, params, device_ptr):
        self._ck(lib.rtx_restir_unpack_state(self._h, C.byref(params), _vp(device_ptr)), "rtx_restir_unpack_state")

    def restir_pack_halo(self, params, halo_px, device_ptr):
        self._ck(lib.rtx_restir_pack_halo(self._h, C.byref(params), halo_px, _vp(device_ptr)), "rtx_restir_pack_halo")

    def restir_unpack_halo(self, params, halo_px, device_ptr):
        self._ck(lib.rtx_restir_unpack_halo(self._h, C.byref(params), halo_px, _vp(device_ptr)), "rtx_restir_unpack_halo")

    def restir_reset(self):
        self._ck(lib.rtx_restir_reset(self._h), "rtx_restir_reset")

    def read_restir_last(self):
        n = lib.rtx_pass1_slots(self.width, self.height)
        di, gi, sd = np.zeros((n, 40), np.uint8), np.zeros((n, 40), np.uint8), np.zeros((n, 60), np.uint8)
        self._ck(lib.rtx_read_restir_last(self._h, _ptr(di), _ptr(gi), _ptr(sd), n), "rtx_read_restir_last")
        return di, gi, sd

    def read_pass1_buffers(self):
        n = lib.rtx_pass1_slots(self.width, self.height)
        di, gi, sd = np.zeros((n, 40), np.uint8), np.zeros((n, 40), np.uint8), np.zeros((n, 60), np.uint8)
        self._ck(lib.rtx_read_pass1_buffers(self._h, _ptr(di), _ptr(gi), _ptr(sd), n), "rtx_read_pass1_buffers")
        return di, gi, sd

    def read_srgb8(self):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._ck(lib.rtx_read_srgb8(self._h, _ptr(out), out.nbytes), "rtx_read_srgb8")
        return out

    def read_layer(self, layer, width=None, height=None):
        """gOutput layer `layer`: 0 = the image, 10-17 = first-hit debug attributes, others black (rtx_read_layer)"""
        w, h = width or self.width, height or self.height
        out = np.zeros((h, w, 4), np.uint8)
        self._ck(lib.rtx_read_layer(self._h, layer, w, h, _ptr(out), out.nbytes), "rtx_read_layer")
        return out

    def stats(self):
        s = Stats()
        self._ck(lib.rtx_get_stats(self._h, C.byref(s)), "rtx_get_stats")
        return s

    def lights(self):
        n = _u32()
        self._ck(lib.rtx_get_lights(self._h, None, 0, C.byref(n)), "rtx_get_lights")
        out = np.zeros((n.value, 20), np.float32)
        if n.value:
            self._ck(lib.rtx_get_lights(self._h, _ptr(out), n.value, C.byref(n)), "rtx_get_lights")
        return out

    def slab_bytes(self, params):
        b = C.c_size_t()
        self._ck(lib.rtx_shard_slab_bytes(C.byref(params), C.byref(b)), "rtx_shard_slab_bytes")
        return b.value

    def pack_tiles(self, params, device_ptr):
        self._ck(lib.rtx_pack_tiles(self._h, C.byref(params), _vp(device_ptr)), "rtx_pack_tiles")

    def unpack_tiles(self, params, device_ptr):
        self.width, self.height = params.width, params.height
        self._ck(lib.rtx_unpack_tiles(self._h, C.byref(params), _vp(device_ptr)), "rtx_unpack_tiles")

    # kernel-level entry points
    def primary_rays(self, params, sample_id=1):
        out = np.zeros((params.height * params.width, 8), np.float32)
        self._ck(lib.rtx_debug_primary_rays(self._h, C.byref(params), sample_id, _ptr(out)), "rtx_debug_primary_rays")
        return out

    def trace_closest(self, rays8):
        r = _f32(rays8).reshape(-1, 8)
        out = np.zeros((len(r), 4), np.float32)
        self._ck(lib.rtx_debug_trace_closest(self._h, _ptr(r), len(r), _ptr(out)), "rtx_debug_trace_closest")
        return out

    def trace_any(self, rays8):
        r = _f32(rays8).reshape(-1, 8)
        out = np.zeros(len(r), np.uint8)
        self._ck(lib.rtx_debug_trace_any(self._h, _ptr(r), len(r), _ptr(out)), "rtx_debug_trace_any")
        return out

    def tree_hash(self):
        """(hash of the node records, hash of the leaf-ordered triangle records) of the wide tree as the device holds it"""
        h = (C.c_uint64 * 2)()
        self._ck(lib.rtx_debug_tree_hash(self._h, h), "rtx_debug_tree_hash")
        return int(h[0]), int(h[1])

    def read_tree(self, which=0):
        """(node records (n, 80) uint8, leaf-ordered triangle records (m, 12) float32) of the device (which=0) or of the host builder's mirror (which=1)"""
        st = self.stats()
        nodes, tris = np.empty((st.bvh_nodes, 80), np.uint8), np.empty((st.bvh_refs, 12), np.float32)
        self._ck(lib.rtx_debug_read_tree(self._h, int(which), nodes.ctypes.data_as(C.c_void_p), C.c_uint64(nodes.nbytes), tris.ctypes.data_as(C.c_void_p), C.c_uint64(tris.nbytes)), "rtx_debug_read_tree")
        return nodes, tris

    def read_host_build(self):
        """(binary tree (n, 16) float32, leaf order (m,) uint32) as the host builder keeps them for refits; empty after a GPU build"""
        nb, lb = C.c_uint64(0), C.c_uint64(0)
        self._ck(lib.rtx_debug_read_host_build(self._h, None, C.byref(nb), None, C.byref(lb)), "rtx_debug_read_host_build")
        nodes, order = np.empty((nb.value // 64, 16), np.float32), np.empty(lb.value // 4, np.uint32)
        self._ck(lib.rtx_debug_read_host_build(self._h, nodes.ctypes.data_as(C.c_void_p), C.byref(nb), order.ctypes.data_as(C.c_void_p), C.byref(lb)), "rtx_debug_read_host_build")
        return nodes, order

    def host_checksums(self):
        """hashes of the host-side scene and build state: (mesh indices, mesh vertices, materials, instances, leaf order, binary tree, wide mirror, shade + objtris + slots)"""
        h = (C.c_uint64 * 8)()
        self._ck(lib.rtx_debug_host_checksums(self._h, h), "rtx_debug_host_checksums")
        return tuple(int(x) for x in h)

    def build_info(self):
        """the last geometry-changing commit: dict(ms=(boxes + keys, sort, PLOC, top on the host, layout), nodes, refs, ploc_rounds, clusters_top); ms all 0 after a host build"""
        ms, cn = (C.c_double * 5)(), (C.c_uint32 * 4)()
        self._ck(lib.rtx_debug_build_info(self._h, ms, cn), "rtx_debug_build_info")
        return dict(ms=tuple(ms), nodes=cn[0], refs=cn[1], ploc_rounds=cn[2], clusters_top=cn[3])

    def trace_counters(self):
        """(node steps, triangle tests) of the closest-hit rays and of the any-hit rays since the last call (OPT_TRACE_COUNTERS 1)"""
        out = (C.c_uint64 * 4)()
        self._ck(lib.rtx_debug_trace_counters(self._h, out), "rtx_debug_trace_counters")
        return tuple(int(v) for v in out)

    def validate_bvh(self):
        """0 when the resident wide BVH (as built, or as refitted on the GPU) covers every triangle inside its decoded boxes"""
        return lib.rtx_debug_validate_bvh(self._h)

    def trace_stats(self, rays8):
        """(n,4): t, node steps, triangle tests, prim bits of a closest-hit BVH traversal"""
        r = _f32(rays8).reshape(-1, 8)
        out = np.zeros((len(r), 4), np.float32)
        self._ck(lib.rtx_debug_trace_stats(self._h, _ptr(r), len(r), _ptr(out)), "rtx_debug_trace_stats")
        return out

    def surface(self, rays8, hits4):
        r, h = _f32(rays8).reshape(-1, 8), _f32(hits4).reshape(-1, 4)
        out = np.zeros((len(r), 16), np.float32)
        self._ck(lib.rtx_debug_surface(self._h, _ptr(r), _ptr(h), len(r), _ptr(out)), "rtx_debug_surface")
        return out

    def bsdf_eval(self, mat_id, flags, n_wo_wi):
        q = _f32(n_wo_wi).reshape(-1, 9)
        out = np.zeros((len(q), 8), np.float32)
        self._ck(lib.rtx_debug_bsdf_eval(self._h, mat_id, flags, _ptr(q), len(q), _ptr(out)), "rtx_debug_bsdf_eval")
        return out

    def bsdf_sample(self, mat_id, flags, n_wo_seed):
        q = _f32(n_wo_seed).reshape(-1, 8)
        out = np.zeros((len(q), 8), np.float32)
        self._ck(lib.rtx_debug_bsdf_sample(self._h, mat_id, flags, _ptr(q), len(q), _ptr(out)), "rtx_debug_bsdf_sample")
        return out

    def tea(self, seed, n):
        s = (C.c_uint32 * 2)(*seed)
        out = np.zeros(n, np.float32)
        self._ck(lib.rtx_debug_tea(self._h, s, n, _ptr(out)), "rtx_debug_tea")
        return out, (s[0], s[1])


class Renderer:
    """The headless Renderer facade (host/Renderer.h) through its C entry points."""

    def __init__(self, width, height, name="rtx", device=0):
        self._h = lib.rtxh_renderer_create(width, height, name.encode(), device)
        self.width, self.height = width, height

    def _ck(self, rc, what):
        if rc != RTX_OK:
            raise RtxError(f"{what} failed: {lib.rtxh_last_error().decode()}")

    def set_scene(self, scene):
        self._ck(lib.rtxh_renderer_set_scene(self._h, scene._h), "set_scene")

    @property
    def params(self):
        return lib.rtxh_renderer_params(self._h).contents

    @property
    def restir_params(self):
        return lib.rtxh_renderer_restir_params(self._h).contents

    def set_mode(self, mode):
        """0 = path tracer (rtx_render), 1 = the reference's ReSTIR frame (rtx_render_restir), one frame per on_render"""
        self._ck(lib.rtxh_renderer_set_mode(self._h, int(mode)), "set_mode")

    def set_option(self, opt, value):
        h = lib.rtxh_renderer_context(self._h)
        if not h or lib.rtx_set_option(h, int(opt), int(value)) != RTX_OK:
            raise RtxError("set_option: " + (lib.rtx_last_error(h).decode() if h else "renderer not initialised"))

    def on_init(self):
        self._ck(lib.rtxh_renderer_on_init(self._h), "OnInit")

    def on_update(self):
        self._ck(lib.rtxh_renderer_on_update(self._h), "OnUpdate")

    def set_instance_transform(self, instance, o2w16):
        """Renderer::SetInstanceTransform: takes effect in the next on_update (transform-only commit = GPU refit)"""
        m = _f32(o2w16).reshape(16)
        self._ck(lib.rtxh_renderer_set_instance_transform(self._h, int(instance), _ptr(m)), "SetInstanceTransform")

    def on_render(self):
        self._ck(lib.rtxh_renderer_on_render(self._h), "OnRender")

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._ck(lib.rtxh_renderer_read_accum(self._h, _ptr(out), out.nbytes), "read_accum")
        return out

    def read_output(self):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._ck(lib.rtxh_renderer_read_output(self._h, _ptr(out), out.nbytes), "read_output")
        return out

    def on_key_up(self, key):
        self._ck(lib.rtxh_renderer_on_key_up(self._h, ord(key) if isinstance(key, str) else key), "OnKeyUp")

    @property
    def display_layer(self):
        return lib.rtxh_renderer_display_layer(self._h)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.rtxh_renderer_destroy(h)

    __del__ = close
