"""legosnark_amd -- MI355X (gfx950) implementation of LegoSNARK's elliptic-curve hot path.

This package is plumbing around the C-ABI shared library (include/legosnark_amd.h):
it loads ``liblegosnark_amd.so`` with ctypes and exposes thin helpers that take numpy
buffers in libff's byte layout or torch CUDA tensors (device pointers).  There is no
CPU fallback: if the library or a gfx950 device is missing every compute call raises.

Host-side C++ users bind the same C-ABI through the libff-compatible header shim in
``legosnark_amd/shim``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# LSA_LIB_VARIANT=loopback: the TEST build whose comm.hip carries the one-GPU stand-in for a multi-rank collective
# (csrc/Makefile); the product library is the default and does not contain it
# LSA_LIB_VARIANT=<name> loads liblegosnark_amd_<name>.so: "loopback" (csrc/Makefile: comm.hip's one-GPU stand-in for a collective,
# tests only) or an experiment build made by hand (e.g. an A/B of a compile-time switch)
_VARIANT = os.environ.get("LSA_LIB_VARIANT", "")
LIB_PATH = os.path.join(_PKG, "liblegosnark_amd_%s.so" % _VARIANT if _VARIANT else "liblegosnark_amd.so")
_lib = None

MSM_STAGES = 8
STAGE_NAMES = ("digits", "scan", "scatter", "accumulate", "reduce", "fold", "order", "total")


class LsaError(RuntimeError):
    pass


class HostStats(C.Structure):
    """lsa_host_stats: host-side wall-clock split of the latest lsa_g1_msm / lsa_g2_msm call."""
    _fields_ = [("n", C.c_size_t), ("cache_hit", C.c_int), ("table", C.c_int), ("h2d_scalars_ms", C.c_double),
                ("fingerprint_wait_ms", C.c_double), ("bases_prepare_ms", C.c_double), ("msm_ms", C.c_double),
                ("total_ms", C.c_double), ("table_building", C.c_int)]


def build(verbose=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-j4"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LsaError("%s is missing: run legosnark_amd.build() / __graft_entry__.build() "
                           "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes
        # in one process cannot both own the GPU, so when torch is installed it must be loaded
        # FIRST: our library's libamdhip64.so.7 dependency then resolves to torch's copy.
        if not os.environ.get("LSA_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.lsa_last_error.restype = C.c_char_p
        L.lsa_stream.restype = C.c_void_p
        L.lsa_bases_size.restype = C.c_size_t
        L.lsa_bases_size.argtypes = [C.c_void_p]
        L.lsa_bases_device_ptr.restype = C.c_void_p
        L.lsa_bases_device_ptr.argtypes = [C.c_void_p]
        L.lsa_bases_has_table.argtypes = [C.c_void_p]
        L.lsa_bases_table_windows.argtypes = [C.c_void_p]
        L.lsa_bases_table_windows.restype = C.c_uint
        L.lsa_msm_field_mults_per_pair.argtypes = [C.c_void_p, C.c_size_t]
        L.lsa_msm_field_mults_per_pair.restype = C.c_uint
        L.lsa_msm_set_table_threshold.argtypes = [C.c_size_t]
        L.lsa_msm_set_table_threshold.restype = None
        L.lsa_bases_destroy.argtypes = [C.c_void_p]
        L.lsa_bases_destroy.restype = None
        L.lsa_msm_window_bits.restype = C.c_uint
        L.lsa_msm_window_bits.argtypes = [C.c_size_t]
        for name in ("lsa_g1_bases_create", "lsa_g2_bases_create"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        for name in ("lsa_g1_msm", "lsa_g2_msm"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.lsa_commit_run_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.lsa_msm_run_segments_async.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("lsa_msm_run", "lsa_msm_run_async"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("lsa_g1_normalize", "lsa_g2_normalize"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("lsa_g1_batch_exp", "lsa_g2_batch_exp"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        for name in ("lsa_g1_sum_async", "lsa_g2_sum_async"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("lsa_g1_sum_on", "lsa_g2_sum_on"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.lsa_stream_join_to.argtypes = [C.c_void_p]
        L.lsa_g1_scalar_mul_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.lsa_g1_sparse_matrix_msm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsa_fr_cppoly_witness.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.lsa_fr_eval_mle.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.lsa_fr_fold.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.lsa_fr_ntt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.lsa_fr_ntt_step.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.lsa_fr_sumcheck_round.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.lsa_fr_scale_upper.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.lsa_fr_eq_table.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int]
        L.lsa_miller_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.lsa_miller_loop_product.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsa_pairing_product.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsa_fq12_product.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsa_final_exponentiation.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.lsa_comm_unique_id.argtypes = [C.c_void_p]
        L.lsa_comm_init.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.lsa_comm_init_file.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.lsa_comm_destroy.restype = None
        L.lsa_shard_range.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.lsa_shard_range.restype = None
        for name in ("lsa_msm_run_sharded", "lsa_msm_run_sharded_async"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("lsa_g1_msm_sharded", "lsa_g2_msm_sharded", "lsa_pairing_product_sharded"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsa_comm_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lsa_pairing_product_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.lsa_g2_precomp_bytes.restype = C.c_size_t
        L.lsa_g2_precompute.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsa_miller_loop_precomp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsa_pairing_terms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.lsa_g2_table_cache.argtypes = [C.c_size_t]
        L.lsa_pairing_set_chunk.argtypes = [C.c_uint]
        L.lsa_g2_table_cache_stats.argtypes = [C.c_void_p]
        L.lsa_crs_cache_configure.argtypes = [C.c_int, C.c_size_t]
        L.lsa_crs_cache_clear.restype = None
        L.lsa_crs_cache_table_after.argtypes = [C.c_uint]
        L.lsa_crs_cache_stats.argtypes = [C.POINTER(C.c_uint64)] * 4
        L.lsa_msm_host_stats.argtypes = [C.POINTER(HostStats)]
        _lib = L
        mapped = rccl_paths()
        if len(mapped) > 1:
            raise LsaError("two RCCL builds are mapped into this process (%s): load torch before anything that links "
                           "/opt/rocm/lib/librccl.so.1, or set LD_LIBRARY_PATH so that both resolve to the same file" % ", ".join(mapped))
    return _lib


def rccl_paths():
    """The librccl files mapped into this process (one per distinct path).  torch's wheel bundles its own librccl.so and
    the library links librccl.so.1: both have the SONAME librccl.so.1, so whichever is loaded first serves both -- lib()
    loads torch first, hence torch's copy.  Exactly ONE must be mapped: two RCCL builds in one process would each own a
    separate set of communicators and IPC handles."""
    lib()
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    paths.add(line.split()[-1])
    except OSError:
        pass
    return sorted(paths)


def _check(rc):
    if rc != 0:
        raise LsaError("legosnark_amd error %d: %s" % (rc, lib().lsa_last_error().decode()))


def init(device=0):
    _check(lib().lsa_init(int(device)))


def shutdown():
    lib().lsa_shutdown()


def device_count():
    return lib().lsa_device_count()


def stream_handle():
    """Raw hipStream_t (int) on which the library launches all kernels."""
    return lib().lsa_stream()


def synchronize():
    _check(lib().lsa_synchronize())


def stream_join():
    """Make the library stream wait for the internally pipelined MSM tails issued so far."""
    _check(lib().lsa_stream_join())


def stream_join_to(stream_handle):
    """Make another HIP stream (raw handle) wait for the MSM tails issued so far; the library
    stream itself keeps running ahead."""
    _check(lib().lsa_stream_join_to(C.c_void_p(int(stream_handle))))


def sum_on(group, d_pts, n, d_out, stream_handle):
    """d_out = sum of n device-resident Jacobian points, on the given HIP stream."""
    fn = lib().lsa_g1_sum_on if group == "g1" else lib().lsa_g2_sum_on
    _check(fn(_ptr(d_pts), n, _ptr(d_out), C.c_void_p(int(stream_handle))))


def _host_ptr(a):
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _ptr(x):
    """numpy array -> host pointer, torch tensor -> data_ptr, int -> as is."""
    if isinstance(x, np.ndarray):
        return _host_ptr(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


_ext_stream = {}


def _after_torch(*tensors):
    """Order the library's HIP stream after torch's current stream when any argument is a torch
    CUDA tensor: the kernels behind these helpers run on lsa_stream(), a non-blocking stream that
    torch's streams are not implicitly ordered with, so inputs still being written by pending
    torch kernels would otherwise be read early.  (Results flow the other way through
    synchronize() / stream_join(); see INTEGRATION.md.)"""
    if not any(hasattr(t, "data_ptr") and getattr(t, "is_cuda", False) for t in tensors):
        return
    import torch
    if torch.cuda.current_stream().query():
        return                       # nothing pending on torch's stream: no event, no cross-stream wait in front of the call
    dev = torch.cuda.current_device()
    h = lib().lsa_stream()
    key = (dev, h)
    ext = _ext_stream.get(key)
    if ext is None:
        ext = _ext_stream[key] = torch.cuda.ExternalStream(h, device=torch.device("cuda", dev))
    ext.wait_stream(torch.cuda.current_stream())


def _group_width(group):
    if group not in ("g1", "g2"):
        raise ValueError(group)
    return 12 if group == "g1" else 24


def msm(group, bases, scalars, chunks=1):
    """multiExpMA drop-in on host buffers (uint64 arrays, libff layout)."""
    w = _group_width(group)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, w)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(len(bases), len(scalars))   # src/utils/globl.h:66
    out = np.zeros(w, dtype=np.uint64)
    fn = lib().lsa_g1_msm if group == "g1" else lib().lsa_g2_msm
    _check(fn(_host_ptr(bases), _host_ptr(scalars), n, chunks, _host_ptr(out)))
    return out


COMM_ID_BYTES = 128


def comm_unique_id():
    """128-byte ncclUniqueId (rank 0 creates it and distributes it to the other ranks)."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(lib().lsa_comm_unique_id(buf))
    return bytes(buf)


def comm_init(rank, world, unique_id):
    """Collective: RCCL communicator over `world` processes, one GPU each."""
    if len(unique_id) != COMM_ID_BYTES:
        raise ValueError("unique id must be %d bytes" % COMM_ID_BYTES)
    buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
    _check(lib().lsa_comm_init(int(rank), int(world), buf))


def comm_init_file(rank, world, path, timeout_s=60):
    _check(lib().lsa_comm_init_file(int(rank), int(world), os.fsencode(path), int(timeout_s)))


def comm_destroy():
    lib().lsa_comm_destroy()


def comm_rank():
    return lib().lsa_comm_rank()


def comm_world():
    return lib().lsa_comm_world()


def comm_join():
    _check(lib().lsa_comm_join())


def shard_range(n, world, rank):
    """libff multi_exp chunk split (the last rank takes the remainder) -- lsa_shard_range."""
    lo, hi = C.c_size_t(), C.c_size_t()
    lib().lsa_shard_range(int(n), int(world), int(rank), C.byref(lo), C.byref(hi))
    return int(lo.value), int(hi.value)


def msm_sharded(group, bases, scalars):
    """SPMD multiExpMA: this rank's slice in (host buffers), the sum over all ranks out."""
    w = _group_width(group)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, w)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(len(bases), len(scalars))
    out = np.zeros(w, dtype=np.uint64)
    fn = lib().lsa_g1_msm_sharded if group == "g1" else lib().lsa_g2_msm_sharded
    _check(fn(_host_ptr(bases), _host_ptr(scalars), n, _host_ptr(out)))
    return out


def pairing_product_sharded(g1, g2):
    """final_exponentiation(prod over all ranks' (P_i, Q_i)); this rank's slice in."""
    g1, g2 = _pairs(g1, g2)
    out = np.zeros(48, dtype=np.uint64)
    _check(lib().lsa_pairing_product_sharded(_host_ptr(g1), _host_ptr(g2), len(g1), _host_ptr(out)))
    return out


CRS_CACHE_OFF, CRS_CACHE_SAMPLED, CRS_CACHE_FULL = 0, 1, 2


def crs_cache_configure(mode, max_bytes=0):
    """Verification mode (CRS_CACHE_*) and device-memory budget of the cache behind msm()."""
    _check(lib().lsa_crs_cache_configure(int(mode), int(max_bytes)))


def crs_cache_clear():
    lib().lsa_crs_cache_clear()


def crs_cache_stats():
    v = [C.c_uint64() for _ in range(4)]
    lib().lsa_crs_cache_stats(*[C.byref(x) for x in v])
    return dict(zip(("hits", "misses", "resident_bytes", "entries"), [int(x.value) for x in v]))


def crs_cache_table_after(hits):
    """Hits before a cached CRS vector's pre-shifted copies are built in the background (0: never)."""
    _check(lib().lsa_crs_cache_table_after(int(hits)))


def crs_cache_wait_tables():
    """Blocks until every background table build has finished and its entry has switched."""
    _check(lib().lsa_crs_cache_wait_tables())


def msm_host_stats():
    st = HostStats()
    _check(lib().lsa_msm_host_stats(C.byref(st)))
    return {k: getattr(st, k) for k, _ in HostStats._fields_}


class Bases:
    """Device-resident, affine-normalised CRS vector (lsa_bases handle)."""

    def __init__(self, group, bases, on_device=False):
        self.group = group
        self.w = _group_width(group)
        h = C.c_void_p()
        if on_device:
            n = bases.numel() * bases.element_size() // (self.w * 8)
            ptr = C.c_void_p(bases.data_ptr())
            _after_torch(bases)
        else:
            bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, self.w)
            n = len(bases)
            ptr = _host_ptr(bases)
        fn = lib().lsa_g1_bases_create if group == "g1" else lib().lsa_g2_bases_create
        _check(fn(ptr, n, 1 if on_device else 0, C.byref(h)))
        self.handle = h
        self.n = n

    def msm(self, d_scalars, n=None, first=0):
        """d_scalars: torch CUDA tensor (n x 32 B Montgomery Fr).  Returns host Jacobian point."""
        if n is None:
            n = self.n - first
        out = np.zeros(self.w, dtype=np.uint64)
        _after_torch(d_scalars)
        _check(lib().lsa_msm_run(self.handle, first, _ptr(d_scalars), n, _host_ptr(out)))
        return out

    def msm_async(self, d_scalars, d_out, n=None, first=0):
        if n is None:
            n = self.n - first
        _after_torch(d_scalars, d_out)
        _check(lib().lsa_msm_run_async(self.handle, first, _ptr(d_scalars), n, _ptr(d_out)))

    def msm_segments_async(self, d_scalars, offsets, d_outs, first=0):
        """len(offsets) - 1 independent MSMs in one pass: result j = sum_i d_scalars[offsets[j] + i] *
        bases[first + i] (prefixes of these bases; CPPoly::prove's ladder).  d_outs: device tensor of
        len(offsets) - 1 points.  Needs has_table()."""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        _after_torch(d_scalars, d_outs)
        _check(lib().lsa_msm_run_segments_async(self.handle, first, _ptr(d_scalars), _host_ptr(off), len(off) - 1, _ptr(d_outs)))

    def msm_sharded_async(self, d_scalars, d_out, n=None, first=0):
        """This rank's slice of a sharded MSM + the RCCL exchange step; d_out (device) receives the
        sum over all ranks.  Needs comm_init(); order results with comm_join() / synchronize()."""
        if n is None:
            n = self.n - first
        _after_torch(d_scalars, d_out)
        _check(lib().lsa_msm_run_sharded_async(self.handle, first, _ptr(d_scalars), n, _ptr(d_out)))

    def msm_sharded(self, d_scalars, n=None, first=0):
        if n is None:
            n = self.n - first
        out = np.zeros(self.w, dtype=np.uint64)
        _after_torch(d_scalars)
        _check(lib().lsa_msm_run_sharded(self.handle, first, _ptr(d_scalars), n, _host_ptr(out)))
        return out

    def device_ptr(self):
        return lib().lsa_bases_device_ptr(self.handle)

    def has_table(self):
        return bool(lib().lsa_bases_has_table(self.handle))

    def table_windows(self):
        return int(lib().lsa_bases_table_windows(self.handle))

    def field_mults_per_pair(self, n=None):
        return int(lib().lsa_msm_field_mults_per_pair(self.handle, self.n if n is None else n))

    def close(self):
        if self.handle:
            lib().lsa_bases_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def commit_async(g1_bases, g2_bases, d_scalars, d_out_g1, d_out_g2, n=None):
    """CommScheme::commit's pair of MSMs (commit.h:154-155) over one device-resident scalar vector:
    d_out_g1 = sum s_i * g1_bases[i], d_out_g2 = sum s_i * g2_bases[i]; one shared scalar sort when
    both handles carry copies over the same number of points."""
    if n is None:
        n = min(g1_bases.n, g2_bases.n)
    _after_torch(d_scalars, d_out_g1, d_out_g2)
    _check(lib().lsa_commit_run_async(g1_bases.handle, g2_bases.handle, _ptr(d_scalars), n, _ptr(d_out_g1), _ptr(d_out_g2)))


def normalize(group, pts):
    w = _group_width(group)
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, w)
    out = np.zeros_like(pts)
    fn = lib().lsa_g1_normalize if group == "g1" else lib().lsa_g2_normalize
    _check(fn(_host_ptr(pts), len(pts), _host_ptr(out)))
    return out


def batch_exp(group, base, scalars, out=None):
    """simpleBatchExp drop-in: out[i] = scalars[i] * base.

    base: host Jacobian point (uint64 limbs).  scalars: numpy (host) or torch CUDA tensor
    (device); `out` must live on the same side (allocated here for the host case)."""
    w = _group_width(group)
    base = np.ascontiguousarray(base, dtype=np.uint64).reshape(w)
    fn = lib().lsa_g1_batch_exp if group == "g1" else lib().lsa_g2_batch_exp
    if isinstance(scalars, np.ndarray):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        n = len(scalars)
        out = np.zeros((n, w), dtype=np.uint64)
        _check(fn(_host_ptr(base), _host_ptr(scalars), n, _host_ptr(out), 0))
        return out
    n = scalars.numel() * scalars.element_size() // 32
    if out is None:
        import torch
        out = torch.empty((n, w), dtype=torch.int64, device=scalars.device)
    _after_torch(scalars, out)
    _check(fn(_host_ptr(base), _ptr(scalars), n, _ptr(out), 1))
    return out


def scalar_mul_batch(pts, scalars, out=None):
    """out[i] = scalars[i] * pts[i] on G1 (variable base).  numpy in -> numpy out (host);
    torch CUDA tensors in -> torch CUDA tensor out (device)."""
    if isinstance(pts, np.ndarray):
        pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 12)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        if len(pts) != len(scalars):
            raise ValueError("need as many scalars as points")
        out = np.zeros_like(pts)
        _check(lib().lsa_g1_scalar_mul_batch(_host_ptr(pts), _host_ptr(scalars), len(pts), _host_ptr(out), 0))
        return out
    n = pts.numel() * pts.element_size() // 96
    if out is None:
        import torch
        out = torch.empty((n, 12), dtype=torch.int64, device=pts.device)
    _after_torch(pts, scalars, out)
    _check(lib().lsa_g1_scalar_mul_batch(_ptr(pts), _ptr(scalars), n, _ptr(out), 1))
    return out


def sparse_matrix_msm(vals, rows, col_ptr, exps):
    """mtxmultiexp on a CSC matrix of G1 elements: out[j] = sum_e exps[rows[e]] * vals[e]."""
    vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, 12)
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    col_ptr = np.ascontiguousarray(col_ptr, dtype=np.uint64)
    exps = np.ascontiguousarray(exps, dtype=np.uint64).reshape(-1, 4)
    ncols = len(col_ptr) - 1
    if ncols < 0 or len(vals) != len(rows) or (ncols >= 0 and int(col_ptr[-1]) != len(vals)):
        raise ValueError("inconsistent CSC arrays")
    out = np.zeros((ncols, 12), dtype=np.uint64)
    _check(lib().lsa_g1_sparse_matrix_msm(_host_ptr(vals), _host_ptr(rows), _host_ptr(col_ptr), ncols,
                                          _host_ptr(exps), len(exps), _host_ptr(out)))
    return out


def _log2_exact(n):
    d = int(n).bit_length() - 1
    if n <= 0 or (1 << d) != n:
        raise ValueError("length must be a power of two")
    return d


def cppoly_witness(v, r, out=None):
    """CPPoly::prove witness coefficients (poly.h:55-67).  numpy (host) or torch CUDA tensors."""
    if isinstance(v, np.ndarray):
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        d = _log2_exact(len(v))
        if len(r) != d:
            raise ValueError("need log2(len(v)) evaluation coordinates")
        w = np.zeros_like(v)
        _check(lib().lsa_fr_cppoly_witness(_host_ptr(v), d, _host_ptr(r), _host_ptr(w), 0))
        return w
    d = _log2_exact(v.numel() * v.element_size() // 32)
    if out is None:
        import torch
        out = torch.empty_like(v)
    _after_torch(v, r, out)
    _check(lib().lsa_fr_cppoly_witness(_ptr(v), d, _ptr(r), _ptr(out), 1))
    return out


def eval_mle(v, r):
    """MultiVPolyT::evalMLE (polytools.h:207-234).  Host numpy in -> (4,) uint64."""
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    d = _log2_exact(len(v))
    if len(r) != d:
        raise ValueError("need log2(len(v)) evaluation coordinates")
    out = np.zeros(4, dtype=np.uint64)
    _check(lib().lsa_fr_eval_mle(_host_ptr(v), d, _host_ptr(r), _host_ptr(out), 0))
    return out


def eval_mle_device(d_v, d_r, d_out):
    d = _log2_exact(d_v.numel() * d_v.element_size() // 32)
    _after_torch(d_v, d_r, d_out)
    _check(lib().lsa_fr_eval_mle(_ptr(d_v), d, _ptr(d_r), _ptr(d_out), 1))
    return d_out


def fr_fold(old, r):
    """DPMle::pushRandomness (mle.h:199-210) on a host vector: returns the folded half."""
    old = np.ascontiguousarray(old, dtype=np.uint64).reshape(-1, 4)
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
    half = len(old) // 2
    cur = np.zeros((half, 4), dtype=np.uint64)
    _check(lib().lsa_fr_fold(_host_ptr(old), half, _host_ptr(r), _host_ptr(cur), 0))
    return cur


def sumcheck_round(tables, suff=None, pre=None, rho_j=None):
    """Coefficients of the sumcheck round polynomial h_j (make_new_h_poly, sumcheck.h:85-106).
    tables: list of m vectors of 2*half Fr (numpy -> host path, torch CUDA tensors -> device
    path); suff: half Fr or None; pre, rho_j: (4,) uint64 or None.  Returns (m+2, 4) uint64
    ((m+1, 4) without the beta factor)."""
    m = len(tables)
    on_device = not isinstance(tables[0], np.ndarray)
    if on_device:
        _after_torch(*tables, suff)
        half = tables[0].numel() * tables[0].element_size() // 64
        ptrs = (C.c_void_p * m)(*[t.data_ptr() for t in tables])
        sp = C.c_void_p(suff.data_ptr()) if suff is not None else None
    else:
        tables = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
        half = len(tables[0]) // 2
        ptrs = (C.c_void_p * m)(*[t.ctypes.data for t in tables])
        if suff is not None:
            suff = np.ascontiguousarray(suff, dtype=np.uint64).reshape(-1, 4)
        sp = _host_ptr(suff) if suff is not None else None
    pre_a = np.ascontiguousarray(pre, dtype=np.uint64).reshape(4) if pre is not None else None
    rho_a = np.ascontiguousarray(rho_j, dtype=np.uint64).reshape(4) if rho_j is not None else None
    out = np.zeros((m + (2 if rho_a is not None else 1), 4), dtype=np.uint64)
    _check(lib().lsa_fr_sumcheck_round(sp, ptrs, m, half, _host_ptr(pre_a) if pre_a is not None else None,
                                       _host_ptr(rho_a) if rho_a is not None else None, _host_ptr(out), 1 if on_device else 0))
    return out


def fr_scale_upper(old, k):
    """DPBeta suffix update on a host vector: cur[p] = old[half + p] * k."""
    old = np.ascontiguousarray(old, dtype=np.uint64).reshape(-1, 4)
    k = np.ascontiguousarray(k, dtype=np.uint64).reshape(4)
    half = len(old) // 2
    cur = np.zeros((half, 4), dtype=np.uint64)
    _check(lib().lsa_fr_scale_upper(_host_ptr(old), half, _host_ptr(k), _host_ptr(cur), 0))
    return cur


def fr_eq_table(r, variant=0):
    """DPBeta::compute_eq_tbl (mle.h:93-105) on a host vector r of d Fr: 2^d entries.  variant 0: as the reference's loop
    computes it (selected by the top index bit only); 1: the eq monomials prod_j eqbit(bit j of p, r[j])."""
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((1 << len(r), 4), dtype=np.uint64)
    _check(lib().lsa_fr_eq_table(_host_ptr(r), len(r), variant, _host_ptr(out), 0))
    return out


def fr_ntt(a, omega, inverse=False, coset=None):
    """libfqfft radix-2 FFT / iFFT / cosetFFT / icosetFFT over Fr.  numpy (host, returns a new
    array) or torch CUDA tensor (device, in place)."""
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    cg = np.ascontiguousarray(coset, dtype=np.uint64).reshape(4) if coset is not None else None
    if isinstance(a, np.ndarray):
        out = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
        log_n = _log2_exact(len(out))
        _check(lib().lsa_fr_ntt(_host_ptr(out), log_n, _host_ptr(omega), 1 if inverse else 0,
                                _host_ptr(cg) if cg is not None else None, 0))
        return out
    log_n = _log2_exact(a.numel() * a.element_size() // 32)
    _after_torch(a)
    _check(lib().lsa_fr_ntt(_ptr(a), log_n, _host_ptr(omega), 1 if inverse else 0, _host_ptr(cg) if cg is not None else None, 1))
    return a


def fr_ntt_step(a, big_log, small_log, omega, inverse=False, coset=None):
    """libfqfft step_radix2_domain FFT / iFFT / cosetFFT / icosetFFT over Fr on 2^big_log + 2^small_log values (omega: a
    primitive 2^(big_log + 1)-th root of unity).  numpy (host, returns a new array) or torch CUDA tensor (device, in place)."""
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    cg = np.ascontiguousarray(coset, dtype=np.uint64).reshape(4) if coset is not None else None
    m = (1 << big_log) + (1 << small_log)
    if isinstance(a, np.ndarray):
        out = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
        if len(out) != m:
            raise ValueError("fr_ntt_step: expected %d values" % m)
        _check(lib().lsa_fr_ntt_step(_host_ptr(out), big_log, small_log, _host_ptr(omega), 1 if inverse else 0,
                                     _host_ptr(cg) if cg is not None else None, 0))
        return out
    if a.numel() * a.element_size() != 32 * m:
        raise ValueError("fr_ntt_step: expected %d values" % m)
    _after_torch(a)
    _check(lib().lsa_fr_ntt_step(_ptr(a), big_log, small_log, _host_ptr(omega), 1 if inverse else 0, _host_ptr(cg) if cg is not None else None, 1))
    return a


def sum_async(group, d_pts, n, d_out):
    """d_out = sum of n device-resident Jacobian points (async on the library stream)."""
    fn = lib().lsa_g1_sum_async if group == "g1" else lib().lsa_g2_sum_async
    _after_torch(d_pts, d_out)
    _check(fn(_ptr(d_pts), n, _ptr(d_out)))


def _pairs(g1, g2):
    g1 = np.ascontiguousarray(g1, dtype=np.uint64).reshape(-1, 12)
    g2 = np.ascontiguousarray(g2, dtype=np.uint64).reshape(-1, 24)
    if len(g1) != len(g2):
        raise ValueError("need as many G1 as G2 points")
    return g1, g2


def miller_loop(g1, g2):
    """out[i] = miller_loop(precompute_G1(P_i), precompute_G2(Q_i)) -> (n, 48) uint64 (Fq12)."""
    g1, g2 = _pairs(g1, g2)
    out = np.zeros((len(g1), 48), dtype=np.uint64)
    _check(lib().lsa_miller_loop(_host_ptr(g1), _host_ptr(g2), len(g1), _host_ptr(out), 0))
    return out


def miller_loop_product(g1, g2):
    """prod_i miller_loop(P_i, Q_i) (n = 2: double_miller_loop) -> (48,) uint64."""
    g1, g2 = _pairs(g1, g2)
    out = np.zeros(48, dtype=np.uint64)
    _check(lib().lsa_miller_loop_product(_host_ptr(g1), _host_ptr(g2), len(g1), _host_ptr(out)))
    return out


def pairing_product(g1, g2):
    """final_exponentiation(prod_i miller_loop(P_i, Q_i)) (n = 1: reduced_pairing) -> (48,) uint64."""
    g1, g2 = _pairs(g1, g2)
    out = np.zeros(48, dtype=np.uint64)
    _check(lib().lsa_pairing_product(_host_ptr(g1), _host_ptr(g2), len(g1), _host_ptr(out)))
    return out


def pairing_product_segments(g1, g2, offsets, final_exp=True):
    """out[j] = final_exponentiation(prod_{i in [offsets[j], offsets[j+1])} miller_loop(P_i, Q_i)) for
    many independent products in one pass (a verifier's shape) -> (nseg, 48) uint64."""
    g1, g2 = _pairs(g1, g2)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    if len(off) < 1 or int(off[-1]) != len(g1):
        raise ValueError("offsets must end at the number of pairs")
    out = np.zeros((len(off) - 1, 48), dtype=np.uint64)
    _check(lib().lsa_pairing_product_segments(_host_ptr(g1), _host_ptr(g2), _host_ptr(off), len(off) - 1, _host_ptr(out),
                                              1 if final_exp else 0))
    return out


G2_PRECOMP_WORDS = (2 + 3 * 102) * 8          # LSA_G2_PRECOMP_BYTES / 8: QX, QY, 102 x {ell_0, ell_VW, ell_VV}


def g2_precompute(g2):
    """libff precompute_G2 for n points -> (n, G2_PRECOMP_WORDS) uint64: QX, QY, then the coefficient
    triples of the 102 ate-loop steps (alt_bn128_ate_G2_precomp as bytes)."""
    g2 = np.ascontiguousarray(g2, dtype=np.uint64).reshape(-1, 24)
    out = np.zeros((len(g2), G2_PRECOMP_WORDS), dtype=np.uint64)
    _check(lib().lsa_g2_precompute(_host_ptr(g2), len(g2), _host_ptr(out)))
    return out


def _precomp_ptrs(tables, index):
    """tables: (m, G2_PRECOMP_WORDS) array; index: per-term row numbers (None: row i) -> ctypes array of row pointers."""
    tables = np.ascontiguousarray(tables, dtype=np.uint64).reshape(-1, G2_PRECOMP_WORDS)
    idx = range(len(tables)) if index is None else [int(i) for i in index]
    base, stride = tables.ctypes.data, G2_PRECOMP_WORDS * 8
    return tables, (C.c_void_p * len(idx))(*[None if i < 0 else base + i * stride for i in idx])


def miller_loop_precomp(g1, tables, index=None):
    """out[i] = miller_loop(precompute_G1(P_i), table of term i) over precomputed G2 values."""
    g1 = np.ascontiguousarray(g1, dtype=np.uint64).reshape(-1, 12)
    tables, ptrs = _precomp_ptrs(tables, index)
    if len(ptrs) != len(g1):
        raise ValueError("need one table per G1 point")
    out = np.zeros((len(g1), 48), dtype=np.uint64)
    _check(lib().lsa_miller_loop_precomp(_host_ptr(g1), ptrs, len(g1), _host_ptr(out)))
    return out


def pairing_terms(g1, offsets, g2=None, tables=None, index=None, flags=None, final_exp=True):
    """out[j] = [final_exponentiation](prod over terms i of segment j of miller_loop(P_i, Q_i), conjugated where
    flags[i] & 1).  Q_i: row index[i] of `tables` (index[i] < 0 or tables None: the point g2[i])."""
    g1 = np.ascontiguousarray(g1, dtype=np.uint64).reshape(-1, 12)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    if len(off) < 1 or int(off[-1]) != len(g1):
        raise ValueError("offsets must end at the number of terms")
    g2p = None
    if g2 is not None:
        g2 = np.ascontiguousarray(g2, dtype=np.uint64).reshape(-1, 24)
        if len(g2) != len(g1):
            raise ValueError("need as many G1 as G2 points")
        g2p = _host_ptr(g2)
    ptrs = None
    if tables is not None:
        tables, ptrs = _precomp_ptrs(tables, index)
        if len(ptrs) != len(g1):
            raise ValueError("need one table index per term")
    fl = None
    if flags is not None:
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        if len(fl) != len(g1):
            raise ValueError("need one flag per term")
    out = np.zeros((len(off) - 1, 48), dtype=np.uint64)
    _check(lib().lsa_pairing_terms(_host_ptr(g1), g2p, ptrs, None if fl is None else _host_ptr(fl), _host_ptr(off), len(off) - 1,
                                   _host_ptr(out), 1 if final_exp else 0))
    return out


def g2_tables_prefetch(g2):
    """Promise the line tables of these G2 points to the device's cache (built when the next Miller call arrives,
    or earlier once LSA_G2_PREFETCH_BATCH = 60 of them are waiting): what the shim's precompute_G2 does."""
    g2 = np.ascontiguousarray(g2, dtype=np.uint64).reshape(-1, 24)
    lib().lsa_g2_tables_prefetch.argtypes = [C.c_void_p, C.c_size_t]
    _check(lib().lsa_g2_tables_prefetch(_host_ptr(g2), len(g2)))


def g2_table_cache(max_tables):
    """Capacity of the device's G2 line-table cache (0: off; clears it)."""
    _check(lib().lsa_g2_table_cache(int(max_tables)))


def pairing_set_chunk(m):
    """Pairs of a product that share one accumulator on the device (0: automatic)."""
    _check(lib().lsa_pairing_set_chunk(int(m)))


def g2_table_cache_stats():
    out = (C.c_uint64 * 4)()
    _check(lib().lsa_g2_table_cache_stats(out))
    return {"hits": out[0], "misses": out[1], "resident": out[2], "evictions": out[3]}


def fq12_product(f):
    """prod_i f[i] -> (48,) uint64 (Fq12 one for an empty batch)."""
    f = np.ascontiguousarray(f, dtype=np.uint64).reshape(-1, 48)
    out = np.zeros(48, dtype=np.uint64)
    _check(lib().lsa_fq12_product(_host_ptr(f), len(f), _host_ptr(out)))
    return out


def final_exponentiation(f):
    f = np.ascontiguousarray(f, dtype=np.uint64).reshape(-1, 48)
    out = np.zeros_like(f)
    _check(lib().lsa_final_exponentiation(_host_ptr(f), len(f), _host_ptr(out), 0))
    return out


def profile_enable(on=True):
    lib().lsa_profile_enable(1 if on else 0)


def profile_last_msm():
    """Average per-stage milliseconds over the MSM calls recorded since profile_enable()."""
    ms = (C.c_float * MSM_STAGES)()
    cnt = lib().lsa_profile_last_msm(ms)
    d = dict(zip(STAGE_NAMES, [float(x) for x in ms]))
    d["calls"] = int(cnt)
    return d


def set_table_threshold(n):
    """Resident bases / MSMs of at least n points use pre-shifted window copies (0 = default 2^19)."""
    lib().lsa_msm_set_table_threshold(int(n))


def msm_window_bits(n):
    return lib().lsa_msm_window_bits(n)
