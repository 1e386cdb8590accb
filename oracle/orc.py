"""orc.py — ctypes binding of the CPU ORACLE (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  Nothing under royaltracer-dx_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


# same source, same results (fmaf is correctly rounded with or without hardware FMA): the -mfma build is only faster
LIB_NAME = "librt_oracle_fma.so" if (_cpu_has_fma() and os.environ.get("ORC_NO_FMA_BUILD", "0") != "1") else "librt_oracle.so"
# ORC_LIB_NAME=librt_oracle_literal.so: the literal-arithmetic variant (oracle/Makefile), loaded only by tests/test_ggx_pins.py in a child process
LIB_NAME = os.environ.get("ORC_LIB_NAME") or LIB_NAME
LIB_PATH = os.path.join(_HERE, LIB_NAME)


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "librt_oracle.so", "librt_oracle_fma.so", "librt_oracle_literal.so"], stdout=subprocess.DEVNULL)


build()
lib = C.CDLL(LIB_PATH)

_vp, _u32 = C.c_void_p, C.c_uint32
_u32p, _fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "width", "height", "spp", "sample_base", "max_bounces", "nee_samples", "rr_start",
        "frame_seed", "flags", "tile_size", "shard_rank", "shard_count")]


def params_from(p):
    """copy any object with the rtx_params fields (e.g. the product's Params) into orc_params"""
    return Params(*[getattr(p, n) for n, _ in Params._fields_])


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)


_sig("orc_create", _vp)
_sig("orc_destroy", None, _vp)
_sig("orc_set_materials", C.c_int, _vp, _vp, _u32)
_sig("orc_add_mesh", C.c_int, _vp, _vp, _u32, _vp, _u32, _vp, _u32p)
_sig("orc_add_instance", C.c_int, _vp, _u32, _fp, _u32p)
_sig("orc_commit", C.c_int, _vp)
_sig("orc_set_instance_transform", C.c_int, _vp, _u32, _fp)
_sig("orc_set_camera", C.c_int, _vp, _fp, _fp)
_sig("orc_set_threads", C.c_int, _vp, C.c_int)
_sig("orc_render", C.c_int, _vp, C.POINTER(Params), _vp, C.POINTER(C.c_uint64))
_sig("orc_srgb8", None, _vp, _u32, _vp)
_sig("orc_render_v6_pass1", C.c_int, _vp, C.POINTER(Params), _vp, _vp, _vp, _vp, C.POINTER(C.c_uint64))
_sig("orc_pass1_slots", C.c_size_t, _u32, _u32)
_sig("orc_restir_frame", C.c_int, _vp, C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint64))
_sig("orc_map_pixel_id", _u32, _u32, _u32, _u32)
_sig("orc_tea", None, _u32p, _u32, _vp)
_sig("orc_seed_init", None, _u32, _u32, _u32, _u32, _u32p)
_sig("orc_sincos", None, C.c_float, _fp, _fp)
_sig("orc_rsqrt", None, _vp, _u32, _vp)
_sig("orc_pow", C.c_float, C.c_float, C.c_float)
_sig("orc_half_round", C.c_float, C.c_float)
_sig("orc_mat4_inverse", None, _fp, _fp)
_sig("orc_primary_rays", C.c_int, _vp, C.POINTER(Params), _u32, _vp)
_sig("orc_trace_closest", C.c_int, _vp, _vp, _u32, C.c_int, _vp)
_sig("orc_trace_any", C.c_int, _vp, _vp, _u32, C.c_int, _vp)
_sig("orc_surface", C.c_int, _vp, _vp, _vp, _u32, _vp)
_sig("orc_num_triangles", _u32, _vp)
_sig("orc_num_lights", _u32, _vp)
_sig("orc_get_lights", C.c_int, _vp, _vp, _u32)
_sig("orc_bsdf_eval", C.c_int, _vp, _u32, _u32, _vp, _u32, _vp)
_sig("orc_bsdf_sample", C.c_int, _vp, _u32, _u32, _vp, _u32, _vp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_vp)


class Oracle:
    def __init__(self):
        self._h = lib.orc_create()

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.orc_destroy(h)

    __del__ = close

    def load(self, scene, aspect):
        """scene: any object with .materials (n,32), .meshes [(verts(n,7), idx, matids)], .instances [(mesh, o2w16)], .view_proj(aspect)"""
        m = _f32(scene.materials).reshape(-1, 32)
        assert lib.orc_set_materials(self._h, _p(m), len(m)) == 0
        for v, i, mid in scene.meshes:
            v, i, mid = _f32(v).reshape(-1, 7), np.ascontiguousarray(i, np.uint32), np.ascontiguousarray(mid, np.uint32)
            rc = lib.orc_add_mesh(self._h, _p(v), len(v), _p(i), len(i), _p(mid), None)
            assert rc == 0, f"orc_add_mesh rc={rc}"
        for mesh, o2w in scene.instances:
            o = _f32(o2w).reshape(16)
            assert lib.orc_add_instance(self._h, mesh, o.ctypes.data_as(_fp), None) == 0
        assert lib.orc_commit(self._h) == 0
        v, p = scene.view_proj(aspect)
        self.set_camera(v, p)
        return self

    def set_instance_transform(self, inst, o2w):
        o = _f32(o2w).reshape(16)
        assert lib.orc_set_instance_transform(self._h, inst, o.ctypes.data_as(_fp)) == 0
        assert lib.orc_commit(self._h) == 0

    def set_camera(self, view, proj):
        v, p = _f32(view).reshape(16), _f32(proj).reshape(16)
        lib.orc_set_camera(self._h, v.ctypes.data_as(_fp), p.ctypes.data_as(_fp))

    def set_threads(self, n):
        lib.orc_set_threads(self._h, n)

    def render(self, params, accum=None):
        p = params_from(params)
        if accum is None:
            accum = np.zeros((p.height, p.width, 4), np.float32)
        cnt = (C.c_uint64 * 3)()
        assert lib.orc_render(self._h, C.byref(p), _p(accum), cnt) == 0
        return accum, (cnt[0], cnt[1], cnt[2])

    def render_v6_pass1(self, params, accum=None):
        """-> accum, (res_di (n,40) u8, res_gi (n,40) u8, sdata (n,60) u8), ray counts"""
        p = params_from(params)
        if accum is None:
            accum = np.zeros((p.height, p.width, 4), np.float32)
        n = lib.orc_pass1_slots(p.width, p.height)
        di, gi, sd = np.zeros((n, 40), np.uint8), np.zeros((n, 40), np.uint8), np.zeros((n, 60), np.uint8)
        cnt = (C.c_uint64 * 3)()
        assert lib.orc_render_v6_pass1(self._h, C.byref(p), _p(accum), _p(di), _p(gi), _p(sd), cnt) == 0
        return accum, (di, gi, sd), (cnt[0], cnt[1], cnt[2])

    def restir_frames(self, params, accum=None, state=None):
        """`params.spp` consecutive ReSTIR frames; state = (cur_di, cur_gi, cur_sd, last_di, last_gi, last_sd) carried between calls"""
        p = params_from(params)
        if accum is None:
            accum = np.zeros((p.height, p.width, 4), np.float32)
        n = lib.orc_pass1_slots(p.width, p.height)
        if state is None:
            state = tuple(np.zeros((n, k), np.uint8) for k in (40, 40, 60, 40, 40, 60))
        tot = np.zeros(3, np.uint64)
        for fr in range(p.spp):
            q = params_from(params); q.spp = 1; q.frame_seed = p.frame_seed + fr
            cnt = (C.c_uint64 * 3)()
            assert lib.orc_restir_frame(self._h, C.byref(q), _p(accum), *[_p(b) for b in state], cnt) == 0
            tot += np.array([cnt[0], cnt[1], cnt[2]], np.uint64)
        return accum, state, tuple(int(v) for v in tot)

    def primary_rays(self, params, sample_id=1):
        p = params_from(params)
        out = np.zeros((p.height * p.width, 8), np.float32)
        lib.orc_primary_rays(self._h, C.byref(p), sample_id, _p(out))
        return out

    def trace_closest(self, rays8, mode=1):
        r = _f32(rays8).reshape(-1, 8)
        out = np.zeros((len(r), 4), np.float32)
        lib.orc_trace_closest(self._h, _p(r), len(r), mode, _p(out))
        return out

    def trace_any(self, rays8, mode=1):
        r = _f32(rays8).reshape(-1, 8)
        out = np.zeros(len(r), np.uint8)
        lib.orc_trace_any(self._h, _p(r), len(r), mode, _p(out))
        return out

    def surface(self, rays8, hits4):
        r, h = _f32(rays8).reshape(-1, 8), _f32(hits4).reshape(-1, 4)
        out = np.zeros((len(r), 16), np.float32)
        lib.orc_surface(self._h, _p(r), _p(h), len(r), _p(out))
        return out

    def lights(self):
        n = lib.orc_num_lights(self._h)
        out = np.zeros((n, 20), np.float32)
        if n:
            lib.orc_get_lights(self._h, _p(out), n)
        return out

    @property
    def num_triangles(self):
        return lib.orc_num_triangles(self._h)

    def bsdf_eval(self, mat_id, flags, n_wo_wi):
        q = _f32(n_wo_wi).reshape(-1, 9)
        out = np.zeros((len(q), 8), np.float32)
        assert lib.orc_bsdf_eval(self._h, mat_id, flags, _p(q), len(q), _p(out)) == 0
        return out

    def bsdf_sample(self, mat_id, flags, n_wo_seed):
        q = _f32(n_wo_seed).reshape(-1, 8)
        out = np.zeros((len(q), 8), np.float32)
        assert lib.orc_bsdf_sample(self._h, mat_id, flags, _p(q), len(q), _p(out)) == 0
        return out


def tea(seed, n):
    s = (C.c_uint32 * 2)(*seed)
    out = np.zeros(n, np.float32)
    lib.orc_tea(s, n, _p(out))
    return out, (s[0], s[1])


def seed_init(x, y, s, frame_seed):
    o = (C.c_uint32 * 2)()
    lib.orc_seed_init(x, y, s, frame_seed, o)
    return o[0], o[1]


def rsqrt(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(len(x), np.float32)
    lib.orc_rsqrt(_p(x), len(x), _p(out))
    return out


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib.orc_sincos(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def pow_(x, y):
    return lib.orc_pow(C.c_float(x), C.c_float(y))


def half_round(x):
    return lib.orc_half_round(C.c_float(x))


def mat4_inverse(m):
    m, o = _f32(m).reshape(16), np.zeros(16, np.float32)
    lib.orc_mat4_inverse(m.ctypes.data_as(_fp), o.ctypes.data_as(_fp))
    return o


def srgb8(accum):
    a = _f32(accum).reshape(-1, 4)
    out = np.zeros((len(a), 4), np.uint8)
    lib.orc_srgb8(_p(a), len(a), _p(out))
    return out.reshape(accum.shape[:-1] + (4,))
