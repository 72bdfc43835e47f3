"""ctypes binding of libgf2bv_hip.so (include/gf2bv_hip.h) for packed inputs.

``LinearSystem`` reaches the solver through the CPython extension ``_internal``
(list-of-int boundary, like the reference).  Synthetic / batch workloads hand over packed
64-bit-word matrices instead -- host numpy arrays or raw device pointers (e.g.
``torch.Tensor.data_ptr()``); that path goes through this module.  No fallback: a missing
library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GF2BV_LIB") or os.path.join(_HERE, "libgf2bv_hip.so")   # env override: kernel experiments

MODE_SINGLE = 0
MODE_AFFINE_SPACE = 1
STATUS_SOLVED = 0
STATUS_INCONSISTENT = 1

# every symbol include/gf2bv_hip.h declares (checked by tests/test_cabi.py)
EXPORTS = [
    "gf2bv_version", "gf2bv_build_id", "gf2bv_device_count", "gf2bv_last_error",
    "gf2bv_solve_digits", "gf2bv_solve_words", "gf2bv_solve_device", "gf2bv_solve_batch_device",
    "gf2bv_solve_batch_digits", "gf2bv_solve_batch_digits_multi",
    "gf2bv_result_status", "gf2bv_result_rank", "gf2bv_result_dimension", "gf2bv_result_words",
    "gf2bv_result_origin", "gf2bv_result_basis", "gf2bv_result_pivots", "gf2bv_result_stats",
    "gf2bv_result_free", "gf2bv_space_combine", "gf2bv_space_open", "gf2bv_space_enumerate", "gf2bv_space_buffer", "gf2bv_space_close",
    "gf2bv_slab_work_words", "gf2bv_slab_tiles", "gf2bv_slab_open", "gf2bv_slab_blocks", "gf2bv_slab_owner",
    "gf2bv_slab_payload_bytes", "gf2bv_slab_factor", "gf2bv_slab_apply", "gf2bv_slab_factor_on", "gf2bv_slab_apply_on",
    "gf2bv_slab_finish_local", "gf2bv_slab_solve",
    "gf2bv_slab_close",
    "gf2bv_synth_device", "gf2bv_residual_device",
    "gf2bv_stream_ceiling_device", "gf2bv_lds_clock_device", "gf2bv_kernel_resources",
    "gf2bv_device_alloc", "gf2bv_device_free", "gf2bv_device_upload", "gf2bv_device_download",
    "gf2bv_pool_trim", "gf2bv_pool_idle_bytes", "gf2bv_host_alloc", "gf2bv_host_free", "gf2bv_host_pool_trim", "gf2bv_plan_gang",
]


class Stats(ctypes.Structure):
    _fields_ = [
        ("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("stride_words", ctypes.c_int64),
        ("rank", ctypes.c_int64), ("dimension", ctypes.c_int64),
        ("status", ctypes.c_int32), ("n_panels", ctypes.c_int32), ("n_sweeps", ctypes.c_int32),
        ("panels_per_sweep", ctypes.c_int32), ("tables_per_sweep", ctypes.c_int32), ("table_bits", ctypes.c_int32),
        ("tile_words", ctypes.c_int32), ("gang_systems", ctypes.c_int32),
        ("sweep_words", ctypes.c_double), ("row_xors", ctypes.c_double),
        ("ms_pack", ctypes.c_float), ("ms_eliminate", ctypes.c_float), ("ms_sweep", ctypes.c_float),
        ("ms_backsub", ctypes.c_float), ("ms_export", ctypes.c_float), ("ms_total", ctypes.c_float),
        ("search_handovers", ctypes.c_int32), ("fast_blocks", ctypes.c_int32),
        ("hbm_words", ctypes.c_double), ("bulk_launches", ctypes.c_int32), ("outer_blocks", ctypes.c_int32),
        ("handover_retries", ctypes.c_int32), ("small_path", ctypes.c_int32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class HipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        pp = ctypes.POINTER(ctypes.c_void_p)
        L.gf2bv_last_error.restype = ctypes.c_char_p
        L.gf2bv_build_id.restype = ctypes.c_char_p
        L.gf2bv_solve_digits.argtypes = [vp, vp, i32, i64, i64, i32, i32, pp]
        L.gf2bv_solve_words.argtypes = [vp, i64, i64, i64, i32, i32, pp]
        L.gf2bv_solve_device.argtypes = [vp, i64, i64, i64, i32, i32, vp, i32, pp]
        L.gf2bv_solve_batch_device.argtypes = [vp, i64, i64, i64, i64, i64, i32, i32, vp, i32, pp]
        L.gf2bv_solve_batch_digits.argtypes = [vp, vp, i32, i64, i64, i64, i32, i32, pp]
        L.gf2bv_solve_batch_digits_multi.argtypes = [vp, vp, i32, i64, i64, i64, i32, vp, i32, pp]
        for name, res in (("gf2bv_result_status", i32), ("gf2bv_result_rank", i64),
                          ("gf2bv_result_dimension", i64), ("gf2bv_result_words", i64)):
            getattr(L, name).restype = res
            getattr(L, name).argtypes = [vp]
        for name in ("gf2bv_result_origin", "gf2bv_result_basis", "gf2bv_result_pivots"):
            getattr(L, name).argtypes = [vp, vp]
        L.gf2bv_result_stats.argtypes = [vp, ctypes.POINTER(Stats)]
        L.gf2bv_result_free.argtypes = [vp]
        L.gf2bv_result_free.restype = None
        L.gf2bv_space_combine.argtypes = [vp, vp, i64, i64, vp, i64, vp]
        L.gf2bv_space_combine.restype = None
        L.gf2bv_space_open.argtypes = [vp, vp, i64, i64, i32, pp]
        L.gf2bv_space_enumerate.argtypes = [vp, ctypes.c_uint64, i64, i32, vp]
        L.gf2bv_space_close.argtypes = [vp]
        L.gf2bv_space_close.restype = None
        L.gf2bv_slab_work_words.argtypes = [i64, i64]
        L.gf2bv_slab_work_words.restype = i64
        L.gf2bv_slab_tiles.argtypes = [i64]
        L.gf2bv_slab_tiles.restype = i64
        L.gf2bv_slab_open.argtypes = [vp, i64, i64, i64, vp, i64, i32, i32, i32, pp]
        L.gf2bv_slab_blocks.argtypes = [vp]
        L.gf2bv_slab_blocks.restype = i64
        L.gf2bv_slab_owner.argtypes = [vp, i32]
        L.gf2bv_slab_payload_bytes.argtypes = [vp]
        L.gf2bv_slab_payload_bytes.restype = i64
        L.gf2bv_slab_factor.argtypes = [vp, i32, vp]
        L.gf2bv_slab_apply.argtypes = [vp, i32, vp]
        L.gf2bv_slab_factor_on.argtypes = [vp, i32, vp, vp]
        L.gf2bv_slab_apply_on.argtypes = [vp, i32, vp, vp]
        L.gf2bv_slab_finish_local.argtypes = [vp]
        L.gf2bv_slab_solve.argtypes = [vp, pp]
        L.gf2bv_slab_close.argtypes = [vp]
        L.gf2bv_slab_close.restype = None
        L.gf2bv_synth_device.argtypes = [vp, i64, i64, i64, ctypes.c_uint64, i32, vp]
        L.gf2bv_residual_device.argtypes = [vp, i64, i64, i64, vp, i32, vp, ctypes.POINTER(i64)]
        L.gf2bv_stream_ceiling_device.argtypes = [i32, i64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.gf2bv_lds_clock_device.argtypes = [i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.gf2bv_kernel_resources.argtypes = [i32, ctypes.POINTER(ctypes.c_int32), i32]
        L.gf2bv_device_alloc.argtypes = [i32, i64, pp]
        L.gf2bv_device_free.argtypes = [i32, vp]
        L.gf2bv_device_upload.argtypes = [i32, vp, vp, i64]
        L.gf2bv_device_download.argtypes = [i32, vp, vp, i64]
        L.gf2bv_host_alloc.argtypes = [i64, pp]
        L.gf2bv_host_free.argtypes = [vp]
        L.gf2bv_host_free.restype = None
        L.gf2bv_host_pool_trim.argtypes = []
        L.gf2bv_host_pool_trim.restype = i64
        L.gf2bv_plan_gang.argtypes = [i64, i64, i64, i64]
        L.gf2bv_plan_gang.restype = i64
        L.gf2bv_pool_trim.argtypes = [i32]
        L.gf2bv_pool_trim.restype = i64
        L.gf2bv_pool_idle_bytes.argtypes = [i32]
        L.gf2bv_pool_idle_bytes.restype = i64
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        msg = lib().gf2bv_last_error().decode(errors="replace")
        if rc == 1:
            raise ValueError(msg)
        raise HipError(f"gf2bv_hip error {rc}: {msg}")


def build_id() -> str:
    """Content hash of the sources the loaded libgf2bv_hip.so was compiled from (gf2bv_amd/build.py)."""
    return lib().gf2bv_build_id().decode()


def device_count() -> int:
    return int(lib().gf2bv_device_count())


@dataclass
class Solution:
    """What one solve returns (the packed-words twin of m4ri_solve's result)."""
    status: int
    rank: int
    dimension: int
    origin: np.ndarray                         # ceil(cols/64) uint64 words, bit j = variable j
    basis: np.ndarray                          # dimension x words (mode 1), else 0 x words
    pivots: np.ndarray                         # column rank profile
    stats: dict = field(default_factory=dict)

    @property
    def solved(self) -> bool:
        return self.status == STATUS_SOLVED

    def origin_int(self) -> int:
        return int.from_bytes(self.origin.tobytes(), "little")

    def basis_ints(self) -> tuple:
        return tuple(int.from_bytes(b.tobytes(), "little") for b in self.basis)


def _take(handle, mode: int) -> Solution:
    L = lib()
    try:
        words = int(L.gf2bv_result_words(handle))
        rank = int(L.gf2bv_result_rank(handle))
        dim = int(L.gf2bv_result_dimension(handle))
        status = int(L.gf2bv_result_status(handle))
        origin = np.zeros(max(words, 1), dtype=np.uint64)
        _check(L.gf2bv_result_origin(handle, origin.ctypes.data))
        nb = dim if (mode == MODE_AFFINE_SPACE and status == STATUS_SOLVED) else 0
        basis = np.zeros((nb, words), dtype=np.uint64)
        if nb:
            _check(L.gf2bv_result_basis(handle, basis.ctypes.data))
        piv = np.zeros(max(rank, 1), dtype=np.int32)
        _check(L.gf2bv_result_pivots(handle, piv.ctypes.data))
        st = Stats()
        _check(L.gf2bv_result_stats(handle, ctypes.byref(st)))
        return Solution(status, rank, dim, origin[:words], basis, piv[:rank].copy(), st.as_dict())
    finally:
        L.gf2bv_result_free(handle)


def solve_words(aug: np.ndarray, rows: int, cols: int, mode: int = MODE_SINGLE, device: int = 0) -> Solution:
    """Solve a packed augmented system held in host memory (rows x stride uint64)."""
    aug = np.ascontiguousarray(aug, dtype=np.uint64)
    stride = aug.shape[1] if aug.ndim == 2 else (cols + 1 + 63) // 64
    h = ctypes.c_void_p()
    _check(lib().gf2bv_solve_words(aug.ctypes.data, rows, cols, stride, mode, device, ctypes.byref(h)))
    return _take(h, mode)


def solve_digits(digits: np.ndarray, offsets: np.ndarray, bits_per_digit: int, rows: int, cols: int,
                 mode: int = MODE_SINGLE, device: int = 0) -> Solution:
    digits = np.ascontiguousarray(digits, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    h = ctypes.c_void_p()
    _check(lib().gf2bv_solve_digits(digits.ctypes.data, offsets.ctypes.data, bits_per_digit, rows, cols, mode,
                                    device, ctypes.byref(h)))
    return _take(h, mode)


def solve_device(d_ptr: int, rows: int, cols: int, stride: int, mode: int = MODE_SINGLE, device: int = 0,
                 stream: int = 0, time_kernels: bool = False) -> Solution:
    """Solve a row-major augmented matrix resident in device memory (the matrix is left untouched)."""
    h = ctypes.c_void_p()
    _check(lib().gf2bv_solve_device(d_ptr, rows, cols, stride, mode, device, stream or None,
                                    1 if time_kernels else 0, ctypes.byref(h)))
    return _take(h, mode)


def solve_batch_device(d_ptr: int, nsys: int, sys_stride: int, rows: int, cols: int, stride: int,
                       mode: int = MODE_SINGLE, device: int = 0, stream: int = 0, time_kernels: bool = False) -> list:
    """nsys equal-shape systems resident in device memory, solved as lock-step gangs.  `stream` = the stream that
    produced the matrices (0 = the null stream): the gangs are ordered after it."""
    hs = (ctypes.c_void_p * max(nsys, 1))()
    rc = lib().gf2bv_solve_batch_device(d_ptr, nsys, sys_stride, rows, cols, stride, mode, device, stream or None,
                                        1 if time_kernels else 0, hs)
    return _take_all(hs, nsys, rc, mode)


def _take_all(hs, nsys: int, rc: int, mode: int) -> list:
    if rc != 0:
        for h in hs:
            if h:
                lib().gf2bv_result_free(h)
        _check(rc)
    return [_take(ctypes.c_void_p(hs[i]), mode) for i in range(nsys)]


def solve_batch_digits(digits: np.ndarray, offsets: np.ndarray, bits_per_digit: int, nsys: int, rows: int,
                       cols: int, mode: int = MODE_SINGLE, device: int = 0, devices=None) -> list:
    """nsys equal-shape systems as digit arrays; offsets has nsys*rows + 1 entries (system-major).
    `devices` (a sequence of device indices, repeats allowed) shards the systems in contiguous blocks over them, one
    host thread per entry (gf2bv_solve_batch_digits_multi); default: everything on `device`."""
    digits = np.ascontiguousarray(digits, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    hs = (ctypes.c_void_p * max(nsys, 1))()
    if devices is None:
        rc = lib().gf2bv_solve_batch_digits(digits.ctypes.data, offsets.ctypes.data, bits_per_digit, nsys, rows, cols,
                                            mode, device, hs)
    else:
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        rc = lib().gf2bv_solve_batch_digits_multi(digits.ctypes.data, offsets.ctypes.data, bits_per_digit, nsys, rows,
                                                  cols, mode, devs.ctypes.data, len(devs), hs)
    return _take_all(hs, nsys, rc, mode)


def solve_batch_words(augs, rows: int, cols: int, mode: int = MODE_SINGLE, device: int = 0) -> list:
    """nsys equal-shape packed systems in host memory ([nsys, rows, stride] uint64): one upload, gang solve."""
    augs = np.ascontiguousarray(augs, dtype=np.uint64)
    nsys, stride = augs.shape[0], augs.shape[2]
    pad = (-stride) % 2                                   # the device entry wants an even stride
    if pad:
        augs = np.concatenate([augs, np.zeros((nsys, rows, pad), dtype=np.uint64)], axis=2)
        stride += pad
    buf = DeviceBuffer(max(augs.nbytes, 16), device)
    try:
        buf.upload(augs)
        return solve_batch_device(buf.ptr, nsys, rows * stride, rows, cols, stride, mode, device)
    finally:
        buf.free()


def space_enumerate(origin: np.ndarray, basis: np.ndarray, first: int, count: int, gray: bool = True,
                    device: int = 0) -> np.ndarray:
    """Elements first .. first+count-1 of the affine space origin + span(basis), materialised on the device
    ([count, words] uint64): Gray order (AffineSpaceIterator) or binary order (AffineSpaceIteratorSlow / get)."""
    origin = np.ascontiguousarray(origin, dtype=np.uint64)
    basis = np.ascontiguousarray(basis, dtype=np.uint64).reshape(-1, len(origin))
    h = ctypes.c_void_p()
    _check(lib().gf2bv_space_open(origin.ctypes.data, basis.ctypes.data, basis.shape[0], len(origin), device, ctypes.byref(h)))
    try:
        out = np.zeros((count, len(origin)), dtype=np.uint64)
        _check(lib().gf2bv_space_enumerate(h, first, count, 1 if gray else 0, out.ctypes.data))
        return out
    finally:
        lib().gf2bv_space_close(h)


def synth_device(d_ptr: int, rows: int, cols: int, stride: int, seed: int, device: int = 0, stream: int = 0):
    _check(lib().gf2bv_synth_device(d_ptr, rows, cols, stride, seed, device, stream or None))


def _mix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def planted_solution(cols: int, seed: int) -> np.ndarray:
    """The solution the synthetic generator plants (gf2bv_synth_device / k_synth: pseudo-row 0xFFFFF of the same counter-based
    generator, RHS = <row, planted>): ceil(cols / 64) words.  A full-rank system has no other solution -- what the bench's
    parity gate and the large-size tests compare solve_one with."""
    cw = (cols + 63) // 64
    s = _mix64(np.array([seed], dtype=np.uint64))[0]
    x = _mix64(s ^ (np.uint64(0xFFFFF << 20) | np.arange(cw, dtype=np.uint64)))
    if cols & 63:
        x[-1] &= np.uint64((1 << (cols & 63)) - 1)
    return x


def residual_device(d_ptr: int, rows: int, cols: int, stride: int, x: np.ndarray, device: int = 0,
                    stream: int = 0) -> int:
    x = np.ascontiguousarray(x, dtype=np.uint64)
    bad = ctypes.c_int64(-1)
    _check(lib().gf2bv_residual_device(d_ptr, rows, cols, stride, x.ctypes.data, device, stream or None,
                                       ctypes.byref(bad)))
    return int(bad.value)


def stream_ceiling(nbytes: int = 2 << 30, device: int = 0) -> dict:
    """Measured streaming rates of this GPU (GB/s): in-place read-XOR-write and read-only."""
    rmw, rd = ctypes.c_double(0), ctypes.c_double(0)
    _check(lib().gf2bv_stream_ceiling_device(device, nbytes, ctypes.byref(rmw), ctypes.byref(rd)))
    return {"rmw_gbs": rmw.value, "read_gbs": rd.value}


def lds_clock(device: int = 0) -> dict:
    """Shader clock (MHz) under an LDS-bound load and the LDS bytes per clock and CU that load reached."""
    mhz, bpc = ctypes.c_double(0), ctypes.c_double(0)
    _check(lib().gf2bv_lds_clock_device(device, ctypes.byref(mhz), ctypes.byref(bpc)))
    return {"shader_mhz": mhz.value, "lds_bytes_per_clk_cu": bpc.value}


def kernel_resources(device: int = 0) -> dict:
    """VGPRs per lane and static LDS bytes of the bulk-update kernel and of the panel kernels that run beside it."""
    out = (ctypes.c_int32 * 20)()
    _check(lib().gf2bv_kernel_resources(device, out, 20))
    names = ("update", "block_fast", "narrow_all", "prio_window", "panel_step")
    res = {nm: {"vgprs": int(out[2 * k]), "lds": int(out[2 * k + 1])} for k, nm in enumerate(names)}
    res["update_outer"] = {"vgprs": int(out[10]), "lds": int(out[11]), "scratch": int(out[12])}      # k_update16k (two-level)
    res["block_fast_narrow"] = {"vgprs": int(out[13]), "lds": int(out[14])}                          # search + narrow step in one launch
    res["block_sparse"] = {"vgprs": int(out[15]), "lds": int(out[16])}                               # k_block_sparse<256, 4> (round 5)
    return res


def pool_trim(device: int = 0) -> int:
    """Return every idle buffer of the library's pool on `device` to the device; bytes freed (gf2bv_pool_trim)."""
    return int(lib().gf2bv_pool_trim(device))


def host_pool_trim() -> int:
    """Free the idle page-locked staging buffers (gf2bv_host_pool_trim); bytes released."""
    return int(lib().gf2bv_host_pool_trim())


def pool_idle_bytes(device: int = 0) -> int:
    """Bytes of idle buffers the pool keeps on `device` right now."""
    return int(lib().gf2bv_pool_idle_bytes(device))


class DeviceBuffer:
    """hipMalloc'd scratch for callers that do not bring torch (tests, plain ctypes users)."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, nbytes
        p = ctypes.c_void_p()
        _check(lib().gf2bv_device_alloc(device, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _check(lib().gf2bv_device_upload(self.device, self.ptr, arr.ctypes.data, arr.nbytes))

    def download(self, dtype=np.uint64) -> np.ndarray:
        out = np.zeros(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib().gf2bv_device_download(self.device, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().gf2bv_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def padded_stride(cols: int, multiple: int = 32) -> int:
    """Row stride (in 64-bit words) for a device-resident system: cols+1 bits, 256-byte rows."""
    wt = (cols + 1 + 63) // 64
    return (wt + multiple - 1) // multiple * multiple
