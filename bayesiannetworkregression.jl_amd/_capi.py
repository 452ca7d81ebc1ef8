"""ctypes binding of include/bnr_hip.h (libbnr_hip.so).  No torch types cross this boundary.

The HIP library is the ONLY compute path of the package: if the shared object is missing or a call fails the
error is raised, never papered over with a CPU fallback."""
import ctypes as C
import os

import numpy as np

from ._build import LIB

BNR_OK, BNR_ERR_BAD_ARG, BNR_ERR_HIP, BNR_ERR_CHOLESKY, BNR_ERR_SAMPLER = 0, 1, 2, 3, 4
BNR_ERR_SAMPLER_CAP = BNR_ERR_SAMPLER
ERRORS = {1: "bad argument", 2: "HIP error", 3: "Cholesky failed after jitter", 4: "sampler attempt cap"}


class BnrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbnr_hip: %s (status %d: %s)" % (msg, code, ERRORS.get(code, "?")))
        self.code = code


class Hyper(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("eta", "zeta", "iota", "aDelta", "bDelta", "nu")]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64)
ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)


class UniqueId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]
_dp = C.c_void_p
_lib = None

_SIGS = {
    "bnr_abi_version": (C.c_int, []),
    "bnr_last_error": (C.c_char_p, []),
    "bnr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bnr_device_synchronize": (C.c_int, [C.c_int32]),
    "bnr_runtime_version": (C.c_int, [C.POINTER(C.c_int)]),
    "bnr_chain_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, C.POINTER(Hyper), C.c_uint64, C.c_int32,
                                   C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bnr_chain_create_typed": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, C.c_int32, _dp, C.POINTER(Hyper), C.c_uint64, C.c_int32,
                                         C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bnr_chain_create_from_matrices": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, _dp, C.POINTER(Hyper),
                                                 C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bnr_chain_create_like": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bnr_chain_destroy": (C.c_int, [C.c_void_p]),
    "bnr_chain_init_prior": (C.c_int, [C.c_void_p]),
    "bnr_chain_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, PROGRESS_CB, C.c_void_p,
                                C.POINTER(C.c_int32)]),
    "bnr_chain_run_async": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "bnr_chain_sync": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "bnr_chain_prepare": (C.c_int, [C.c_void_p]),
    "bnr_group_prepare": (C.c_int, [C.c_void_p]),
    "bnr_group_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "bnr_group_destroy": (C.c_int, [C.c_void_p]),
    "bnr_group_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, PROGRESS_CB, C.c_void_p,
                                C.POINTER(C.c_int32)]),
    "bnr_group_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "bnr_group_last_timing": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "bnr_gibbs_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "bnr_chain_get_iter": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "bnr_chain_set_iter": (C.c_int, [C.c_void_p, C.c_int64]),
    "bnr_chain_fetch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [_dp] * 11),
    "bnr_chain_load": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [_dp] * 11),
    "bnr_chain_move_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "bnr_chain_resize": (C.c_int, [C.c_void_p, C.c_int32]),
    "bnr_chain_rhat_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "bnr_chain_summary": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]),
    "bnr_comm_unique_id": (C.c_int, [C.POINTER(UniqueId)]),
    "bnr_comm_create_rccl": (C.c_int, [C.POINTER(UniqueId), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bnr_comm_create_callback": (C.c_int, [C.c_int32, C.c_int32, ALLGATHER_CB, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bnr_comm_destroy": (C.c_int, [C.c_void_p]),
    "bnr_comm_allgather": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64]),
    "bnr_comm_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 5),
    "bnr_rhat": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, _dp, _dp]),
    "bnr_rhat_from_stats": (C.c_int, [_dp, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "bnr_chain_ess_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "bnr_ess_from_stats": (C.c_int, [_dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "bnr_chain_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "bnr_chain_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "bnr_chain_last_timing": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "bnr_chain_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "bnr_chain_debug_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]),
    "bnr_chain_debug_copy": (C.c_int, [C.c_void_p, C.c_int32, _dp, C.c_int64]),
    "bnr_chain_debug_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "bnr_chain_debug_time_gram": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "bnr_debug_set_exp": (C.c_int, [C.c_int32, C.c_int32]),
    "bnr_host_philox": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bnr_host_uniform2": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "bnr_host_normal": (C.c_double, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "bnr_host_gamma": (C.c_double, [C.c_uint64, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]),
    "bnr_host_gig": (C.c_double, [C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32]),
    "bnr_host_edge_index": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "bnr_host_gram_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
}
for _u in ("tau2", "u_xi", "gamma", "D", "theta", "Delta", "M", "mu", "Lambda", "pi"):
    _SIGS["bnr_update_" + _u] = (C.c_int, [C.c_void_p, C.c_int32, C.c_int64])

EXPORTS = sorted(_SIGS)


_foreign_hip = None      # set by lib(): why chains must not be created in this process (GPU-free entry points stay usable)


def _mapped(needle):
    """paths of the shared objects in this process whose name contains `needle` (from /proc/self/maps)"""
    out = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 6 and needle in os.path.basename(parts[5]) and parts[5] not in out:
                    out.append(parts[5])
    except OSError:
        pass
    return out


def lib():
    """Load libbnr_hip.so; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError("libbnr_hip.so is missing at %s -- build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB)
        # Load order: the process serves every libamdhip64 user from the FIRST copy that was mapped.  A torch wheel carries its own
        # ROCm runtime; if it is already in the process, libbnr_hip.so would silently run on that one (its graph capture of the
        # two-branch sweep has only ever been validated on the runtime the library was built against, and round 2 saw captures
        # crash under the wheel's 7.0 runtime).  Refuse with a clear message instead (BNR_ALLOW_FOREIGN_HIP=1 overrides).
        before = _mapped("libamdhip64")
        L = C.CDLL(LIB)
        after = _mapped("libamdhip64")
        foreign = [p_ for p_ in before if not os.path.realpath(p_).startswith(os.path.realpath(os.environ.get("ROCM_PATH", "/opt/rocm")))]
        global _foreign_hip
        if foreign and after == before and not os.environ.get("BNR_ALLOW_FOREIGN_HIP"):
            _foreign_hip = ("libbnr_hip.so was loaded after another HIP runtime (%s) and is bound to it instead of %s/lib: load the library "
                            "(bnr_amd._capi.lib(), or create the chains) BEFORE importing torch, or set BNR_ALLOW_FOREIGN_HIP=1 to run on the "
                            "foreign runtime" % (", ".join(foreign), os.environ.get("ROCM_PATH", "/opt/rocm")))
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != BNR_OK:
        raise BnrError(code, lib().bnr_last_error().decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


TABLE_COLUMNS = ["tau2", "u", "xi", "gamma", "S", "theta", "Delta", "M", "mu", "lam", "pi"]
DEAD_COLUMNS = ["Sigma_inv", "invC", "mu_t"]   # allocated, never written by the reference (gibbs.jl:840-841)


def table_shapes(V, R):
    q = V * (V + 1) // 2
    return dict(tau2=(1, 1), u=(R, V), xi=(V, 1), gamma=(q, 1), S=(q, 1), theta=(1, 1), Delta=(1, 1), M=(R, R),
                mu=(1, 1), lam=(R, 1), pi=(R, 3), Sigma_inv=(R, R), invC=(R, R), mu_t=(R, 1))


def new_table(tot, V, R, dead=True):
    """The reference's state Table: every column Array{Float64,3}(tot,d1,d2), iteration index fastest."""
    names = TABLE_COLUMNS + (DEAD_COLUMNS if dead else [])
    sh = table_shapes(V, R)
    return {k: np.zeros((tot,) + sh[k], dtype=np.float64, order="F") for k in names}


def device_count():
    n = C.c_int(0)
    check(lib().bnr_device_count(C.byref(n)))
    return n.value


def device_synchronize(device=0):
    check(lib().bnr_device_synchronize(int(device)))


def runtime_version():
    v = C.c_int(0)
    check(lib().bnr_runtime_version(C.byref(v)))
    return v.value


X_DTYPES = {np.dtype(np.float64): 0, np.dtype(np.bool_): 1, np.dtype(np.uint8): 1, np.dtype(np.int32): 2, np.dtype(np.int64): 3,
            np.dtype(np.float32): 4}     # the enum of include/bnr_hip.h (BNR_F64, BNR_U8, BNR_I32, BNR_I64, BNR_F32)


class XInput:
    """The model input of a fit as the library takes it: the n x q matrix X_new in its own element type (gibbs.jl:917:
    Matrix{eltype(T)} -- Bool adjacency data stays one byte per entry until it is on the device), or, with x_transform=True, the
    list of n adjacency matrices themselves (setup_X!, gibbs.jl:239-247, then runs on the device).  Element types without a device
    converter are promoted to float64 here."""

    def __init__(self, X, x_transform=False):
        self.from_matrices = bool(x_transform)
        if x_transform:
            mats = [np.asarray(m) for m in X]
            V = mats[0].shape[0]
            if any(m.shape != (V, V) for m in mats):
                raise ValueError("every adjacency matrix must be V x V")
            dt = np.result_type(*mats)
            dt = dt if dt in X_DTYPES else np.dtype(np.float64)
            self.data = [np.asfortranarray(m, dtype=dt) for m in mats]
            self.n, self.V = len(mats), V
            self.q = V * (V + 1) // 2
        else:
            a = np.asarray(X)
            if a.ndim != 2:
                raise ValueError("X must be an n x q matrix")
            dt = a.dtype if a.dtype in X_DTYPES else np.dtype(np.float64)
            self.data = np.asfortranarray(a, dtype=dt)
            self.n, self.q = a.shape
            self.V = int(round((-1 + np.sqrt(1 + 8 * self.q)) / 2))
            if self.V * (self.V + 1) // 2 != self.q:
                raise ValueError("X must have V(V+1)/2 columns")
        self.dtype_code = X_DTYPES[dt]


class Chain:
    """One Gibbs chain resident on one GPU (handle of include/bnr_hip.h)."""

    def __init__(self, X, y, R, tot_save, seed, chain_id, device=0, eta=1.01, zeta=1.0, iota=1.0, aDelta=1.0,
                 bDelta=1.0, nu=10):
        xi = X if isinstance(X, XInput) else XInput(X)
        yf = np.ascontiguousarray(y, dtype=np.float64)
        n, q, V = xi.n, xi.q, xi.V
        if yf.shape != (n,):
            raise ValueError("y must have one entry per row of X")
        self.n, self.q, self.V, self.R, self.tot = n, q, V, int(R), int(tot_save)
        self.h = C.c_void_p()
        hy = Hyper(eta, zeta, iota, aDelta, bDelta, float(nu))
        self.L = lib()
        if _foreign_hip:
            raise BnrError(BNR_ERR_HIP, _foreign_hip)
        common = (_ptr(yf), C.byref(hy), C.c_uint64(int(seed) & (2**64 - 1)), int(chain_id), int(device), int(tot_save), C.byref(self.h))
        if xi.from_matrices:
            ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in xi.data])
            check(self.L.bnr_chain_create_from_matrices(n, V, int(R), ptrs, xi.dtype_code, *common))
        elif xi.dtype_code == 0:
            check(self.L.bnr_chain_create(n, V, int(R), _ptr(xi.data), *common))
        else:
            check(self.L.bnr_chain_create_typed(n, V, int(R), _ptr(xi.data), xi.dtype_code, *common))

    @classmethod
    def like(cls, donor, seed, chain_id, tot_save=None):
        """Another chain of the same fit (same X, y, hyper-parameters; device inputs shared with `donor`)."""
        self = cls.__new__(cls)
        self.n, self.q, self.V, self.R = donor.n, donor.q, donor.V, donor.R
        self.tot = int(donor.tot if tot_save is None else tot_save)
        self.h = C.c_void_p()
        self.L = donor.L
        check(self.L.bnr_chain_create_like(donor.h, C.c_uint64(int(seed) & (2**64 - 1)), int(chain_id), self.tot, C.byref(self.h)))
        return self

    def close(self):
        h, self.h = getattr(self, "h", None), None          # (no ctypes call when the interpreter is shutting down: C may be gone)
        if h and h.value and getattr(self, "L", None) is not None:
            self.L.bnr_chain_destroy(h)

    __del__ = close

    def init_prior(self):
        check(self.L.bnr_chain_init_prior(self.h))

    def run(self, first_index, nburn, total, purge_burn=None, prog_freq=0, callback=None):
        nxt = C.c_int32(0)
        cb = PROGRESS_CB(lambda user, done: callback(done)) if callback else PROGRESS_CB()
        check(self.L.bnr_chain_run(self.h, first_index, nburn, total, purge_burn or 0, prog_freq if callback else 0, cb,
                                   None, C.byref(nxt)))
        return nxt.value

    def prepare(self):
        """Capture the hipGraphs the run loop replays now (otherwise done lazily by the first run call)."""
        check(self.L.bnr_chain_prepare(self.h))

    def run_async(self, first_index, nburn, total, purge_burn=None):
        check(self.L.bnr_chain_run_async(self.h, first_index, nburn, total, purge_burn or 0))

    def sync(self):
        nxt = C.c_int32(0)
        check(self.L.bnr_chain_sync(self.h, C.byref(nxt)))
        return nxt.value

    def gibbs_step(self, row, it):
        check(self.L.bnr_gibbs_step(self.h, row, it))

    def update(self, name, row, it):
        check(getattr(self.L, "bnr_update_" + name)(self.h, row, it))

    @property
    def iter(self):
        v = C.c_int64(0)
        check(self.L.bnr_chain_get_iter(self.h, C.byref(v)))
        return v.value

    @iter.setter
    def iter(self, v):
        check(self.L.bnr_chain_set_iter(self.h, int(v)))

    def fetch(self, first_row=1, last_row=None, table=None, host_row_offset=0):
        last_row = self.tot if last_row is None else last_row
        if table is None:
            table = new_table(last_row - first_row + 1, self.V, self.R, dead=False)
            host_row_offset = -(first_row - 1)
        tot = table["tau2"].shape[0]
        for k in TABLE_COLUMNS:
            a = table[k]
            assert a.dtype == np.float64 and a.flags.f_contiguous and a.shape[0] == tot, k
        check(self.L.bnr_chain_fetch(self.h, first_row, last_row, tot, host_row_offset, *[_ptr(table[k]) for k in TABLE_COLUMNS]))
        return table

    def load(self, table, first_row=1, last_row=None, host_row_offset=0):
        tot = table["tau2"].shape[0]
        last_row = min(self.tot, tot - host_row_offset) if last_row is None else last_row
        cols = []
        for k in TABLE_COLUMNS:
            a = table.get(k)
            if a is not None:
                a = np.asfortranarray(a, dtype=np.float64)
                assert a.shape[0] == tot, k
            cols.append(a)
        self._keep = cols
        check(self.L.bnr_chain_load(self.h, first_row, last_row, tot, host_row_offset, *[_ptr(a) for a in cols]))

    def move_rows(self, to_row, from_row, count):
        check(self.L.bnr_chain_move_rows(self.h, to_row, from_row, count))

    def resize(self, new_tot):
        check(self.L.bnr_chain_resize(self.h, new_tot))
        self.tot = int(new_tot)

    def rhat_stats(self, first_row, nsamp):
        out = np.empty(4 * (self.q + self.V))
        check(self.L.bnr_chain_rhat_stats(self.h, first_row, nsamp, _ptr(out)))
        return out

    def summary(self, first_row, nsamp, k_lo, k_hi):
        """(mean gamma, k_lo-th smallest, k_hi-th smallest per edge, mean xi per node) over the row window, on the device."""
        mean, lo, hi, pxi = np.empty(self.q), np.empty(self.q), np.empty(self.q), np.empty(self.V)
        check(self.L.bnr_chain_summary(self.h, first_row, nsamp, k_lo, k_hi, _ptr(mean), _ptr(lo), _ptr(hi), _ptr(pxi)))
        return mean, lo, hi, pxi

    def ess_stats(self, first_row, nsamp, max_lag):
        out = np.empty(2 * (2 + max_lag) * (self.q + self.V))
        check(self.L.bnr_chain_ess_stats(self.h, first_row, nsamp, max_lag, _ptr(out)))
        return out

    def counters(self):
        out = (C.c_int64 * 8)()
        check(self.L.bnr_chain_counters(self.h, out))
        return dict(jitter=out[0], nan_w=out[1], sampler_cap=out[2], chol_fail=out[3], where=list(out[4:8]))

    def set_profiling(self, on=True):
        check(self.L.bnr_chain_set_profiling(self.h, 1 if on else 0))

    def last_timing(self, which):
        us, n = C.c_double(0), C.c_int64(0)
        check(self.L.bnr_chain_last_timing(self.h, which, C.byref(us), C.byref(n)))
        return us.value, n.value

    def debug_read(self, count=1024):
        out = (C.c_uint64 * count)()
        check(self.L.bnr_chain_debug_read(self.h, out, count))
        return np.array(out[:], dtype=np.uint64)

    def debug_time_gram(self, reps=100):
        us = C.c_double(0)
        check(self.L.bnr_chain_debug_time_gram(self.h, reps, C.byref(us)))
        return us.value

    def debug_copy(self, which, count):
        out = np.empty(count)
        check(self.L.bnr_chain_debug_copy(self.h, which, _ptr(out), count))
        return out

    def debug_dims(self):
        out = (C.c_int32 * 8)()
        check(self.L.bnr_chain_debug_dims(self.h, out))
        return dict(zip(("n_pad", "q_pad", "ksplit", "ntile", "kcp", "kslab", "i8L", "rowlen"), [int(v) for v in out]))

    def debug_gram(self):
        """The (G) of the last Gram launch, lower triangle, summed over the K-split partial tiles on the host (diagnostics; n_pad x n_pad)."""
        dm = self.debug_dims()
        nt, ks = dm["ntile"], dm["ksplit"]
        ntl = nt * (nt + 1) // 2
        P = self.debug_copy(3, ks * ntl * 4096).reshape(ks, ntl, 64, 64).sum(axis=0)     # [tile][j][i]
        G = np.zeros((dm["n_pad"], dm["n_pad"]))
        for ti in range(nt):
            for tj in range(ti + 1):
                G[ti * 64:(ti + 1) * 64, tj * 64:(tj + 1) * 64] = P[ti * (ti + 1) // 2 + tj].T
        return np.tril(G)

    def set_option(self, name, value):
        check(self.L.bnr_chain_set_option(self.h, name.encode(), int(value)))


class Group:
    """Lockstep group of equally shaped chains on one GPU (bnr_group_* of include/bnr_hip.h): one launch per kernel of a
    sweep for all members; every member's table is bitwise what it would be when run alone."""

    def __init__(self, chains):
        self.chains = list(chains)
        self.L = lib()
        self.h = C.c_void_p()
        arr = (C.c_void_p * len(self.chains))(*[ch.h for ch in self.chains])
        check(self.L.bnr_group_create(arr, len(self.chains), C.byref(self.h)))

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h and h.value and getattr(self, "L", None) is not None:
            self.L.bnr_group_destroy(h)

    __del__ = close

    def run(self, first_index, nburn, total, purge_burn=None, prog_freq=0, callback=None):
        nxt = C.c_int32(0)
        cb = PROGRESS_CB(lambda user, done: callback(done)) if callback else PROGRESS_CB()
        check(self.L.bnr_group_run(self.h, first_index, nburn, total, purge_burn or 0, prog_freq if callback else 0, cb,
                                   None, C.byref(nxt)))
        return nxt.value

    def prepare(self):
        check(self.L.bnr_group_prepare(self.h))

    def set_option(self, name, value):
        check(self.L.bnr_group_set_option(self.h, name.encode(), int(value)))

    def set_profiling(self, on=True):
        self.set_option("profiling", 1 if on else 0)

    def last_timing(self, which):
        us, n = C.c_double(0), C.c_int64(0)
        check(self.L.bnr_group_last_timing(self.h, which, C.byref(us), C.byref(n)))
        return us.value, n.value


class Comm:
    """The ranks of one fit (bnr_comm of include/bnr_hip.h): RCCL communicator owned by the library, or a host callback."""

    def __init__(self, handle, rank, world, keep=None):
        self.h, self.rank, self.world, self._keep, self.L = handle, rank, world, keep, lib()

    @staticmethod
    def unique_id():
        uid = UniqueId()
        check(lib().bnr_comm_unique_id(C.byref(uid)))
        return bytes(C.string_at(C.addressof(uid), 128))

    @classmethod
    def rccl(cls, unique_id, rank, world, device):
        """Collective: every rank calls it with rank 0's 128-byte id."""
        uid = UniqueId()
        C.memmove(C.addressof(uid), unique_id, 128)
        h = C.c_void_p()
        check(lib().bnr_comm_create_rccl(C.byref(uid), rank, world, device, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def callback(cls, rank, world, allgather):
        """allgather(send: ndarray[count]) -> ndarray[world * count] in rank order (the host's own transport)."""
        def _cb(_ctx, send, recv, count):
            try:
                out = np.ascontiguousarray(allgather(np.ctypeslib.as_array(send, shape=(count,)).copy()), dtype=np.float64).reshape(-1)
                if out.size != world * count:
                    return 2
                C.memmove(recv, out.ctypes.data, 8 * out.size)
                return 0
            except Exception:                                                      # no exception may cross the ABI
                import traceback
                traceback.print_exc()
                return 1
        fn = ALLGATHER_CB(_cb)
        h = C.c_void_p()
        check(lib().bnr_comm_create_callback(rank, world, fn, None, C.byref(h)))
        return cls(h, rank, world, keep=fn)

    def allgather(self, send):
        s = np.ascontiguousarray(send, dtype=np.float64).reshape(-1)
        out = np.empty(self.world * s.size)
        check(self.L.bnr_comm_allgather(self.h, _ptr(s), _ptr(out), s.size))
        return out.reshape(self.world, s.size)

    def info(self):
        """bnr_comm_info: what the transport itself reports -- {'kind': 'rccl' | 'callback', 'rank', 'world', 'rccl_ranks' (ncclCommCount; 0 unless RCCL), 'rccl_rank'}."""
        v = [C.c_int32() for _ in range(5)]
        check(self.L.bnr_comm_info(self.h, *[C.byref(x) for x in v]))
        kind, rank, world, nr, ur = [x.value for x in v]
        return {"kind": {0: "none", 1: "rccl", 2: "callback"}[kind], "rank": rank, "world": world, "rccl_ranks": nr, "rccl_rank": ur}

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h and h.value and getattr(self, "L", None) is not None:
            self.L.bnr_comm_destroy(h)

    __del__ = close


def rhat(chains, nchains_total, comm, burn, nsamp, V=None, q=None):
    """bnr_rhat: split-Rhat of gamma and xi over ALL chains of the fit; `chains` = this rank's Chain objects in increasing
    chain id.  Returns (rhat_gamma[q], rhat_xi[V]).  V, q are needed on a rank that holds no chain."""
    if chains:
        V, q = chains[0].V, chains[0].q
    rx, rg = np.empty(V), np.empty(q)
    arr = (C.c_void_p * max(1, len(chains)))(*[ch.h for ch in chains])
    check(lib().bnr_rhat(arr, len(chains), nchains_total, comm.h if comm is not None else None, burn, nsamp, _ptr(rx), _ptr(rg)))
    return rg, rx


def ess_from_stats(stats, nsamp, max_lag):
    """Bulk effective sample size from the gathered per-chain messages (nchains, 2 * (2 + max_lag) * nparams)."""
    st = np.ascontiguousarray(stats, dtype=np.float64)
    nch = st.shape[0]
    npar = st.shape[1] // (2 * (2 + max_lag))
    out = np.empty(npar)
    check(lib().bnr_ess_from_stats(_ptr(st), nch, npar, nsamp, max_lag, _ptr(out)))
    return out


def rhat_from_stats(stats, nsamp):
    """stats: (nchains, 4*nparams) -> rhat (nparams,)   second half of convergence.jl:4-65."""
    s = np.ascontiguousarray(stats, dtype=np.float64)
    nchains, w = s.shape
    out = np.empty(w // 4)
    check(lib().bnr_rhat_from_stats(_ptr(s), nchains, w // 4, nsamp, _ptr(out)))
    return out
