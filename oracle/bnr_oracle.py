"""ctypes binding of the CPU oracle (oracle/bnr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("BNR_ORACLE_LIB") or os.path.join(_HERE, "libbnr_oracle.so")     # BNR_ORACLE_LIB: the sanitizer build (tools/sanitize_cpu.sh)

SITES = dict(INIT_S=1, INIT_PI=2, INIT_LAM=3, INIT_XI=4, INIT_M_CHI=5, INIT_M_N=6, INIT_U=7, INIT_GAMMA=8,
             TAU2=16, XI=17, U_Z=18, G_Z1=19, G_Z2=20, D_GIG=21, D_GAMMA=22, THETA=23, DELTA=24,
             DELTA_COIN=25, M_CHI=26, M_N=27, MU=28, LAMBDA=29, PI=30)

COLUMNS = ["tau2", "u", "xi", "gamma", "S", "theta", "Delta", "M", "mu", "lam", "pi"]


def build(force=False):
    src = os.path.join(_HERE, "bnr_oracle.c")
    if os.environ.get("BNR_ORACLE_LIB"):
        return _LIB
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbnr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class _Orc(C.Structure):
    _fields_ = [("n", C.c_int32), ("V", C.c_int32), ("R", C.c_int32), ("q", C.c_int32),
                ("eta", C.c_double), ("zeta", C.c_double), ("iota", C.c_double),
                ("aDelta", C.c_double), ("bDelta", C.c_double), ("nu", C.c_double),
                ("seed", C.c_uint64), ("X", C.c_void_p), ("y", C.c_void_p),
                ("pdf_mode", C.c_int32), ("cost_mode", C.c_int32), ("tot", C.c_int32), ("status", C.c_int32),
                ("tau2", C.c_void_p), ("u", C.c_void_p), ("xi", C.c_void_p), ("gamma", C.c_void_p),
                ("S", C.c_void_p), ("theta", C.c_void_p), ("Delta", C.c_void_p), ("M", C.c_void_p),
                ("mu", C.c_void_p), ("lam", C.c_void_p), ("pi", C.c_void_p),
                ("iter", C.c_int64), ("jitter_events", C.c_int64), ("nan_w_events", C.c_int64),
                ("gig_branch", C.c_int64 * 5)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        assert _lib.orc_sizeof() == C.sizeof(_Orc), (_lib.orc_sizeof(), C.sizeof(_Orc))
        _lib.orc_normal.restype = C.c_double
        _lib.orc_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.orc_uniform2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib.orc_gamma.restype = C.c_double
        _lib.orc_gamma.argtypes = [C.c_void_p, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.orc_sample_gig.restype = C.c_double
        _lib.orc_sample_gig.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32]
        _lib.orc_run.restype = C.c_int
        _lib.orc_set_blas.argtypes = [C.c_void_p, C.c_void_p]
    return _lib


_blas = None


def use_numpy_openblas(threads=None):
    """Back the reference-cost timing mode (cost_mode=2) with the OpenBLAS that ships inside numpy's wheel (numpy.libs/
    libscipy_openblas64_*.so: ILP64 Fortran symbols scipy_dgemm_64_ / scipy_dgesv_64_) -- the image has no system BLAS.
    Returns a description string, or None when the library or a symbol is missing (cost_mode 2 then behaves like 1)."""
    global _blas
    import glob
    import numpy
    if _blas is None:
        cand = glob.glob(os.path.join(os.path.dirname(numpy.__file__), "..", "numpy.libs", "libscipy_openblas*.so"))
        for path in cand:
            try:
                B = C.CDLL(path)
                gemm, gesv = C.cast(B.scipy_dgemm_64_, C.c_void_p), C.cast(B.scipy_dgesv_64_, C.c_void_p)
            except (OSError, AttributeError):
                continue
            lib().orc_set_blas(gemm, gesv)
            _blas = (B, os.path.basename(path))
            break
    if _blas is None:
        return None
    B, name = _blas
    if threads is not None:
        B.scipy_openblas_set_num_threads64_(C.c_int(int(threads)))
    B.scipy_openblas_get_config64_.restype = C.c_char_p
    return "%s (%s), %d threads" % (name, B.scipy_openblas_get_config64_().decode().strip(), B.scipy_openblas_get_num_threads64_())


def table_shapes(V, R):
    q = V * (V + 1) // 2
    return dict(tau2=(1, 1), u=(R, V), xi=(V, 1), gamma=(q, 1), S=(q, 1), theta=(1, 1), Delta=(1, 1),
                M=(R, R), mu=(1, 1), lam=(R, 1), pi=(R, 3))


def new_table(tot, V, R, fill=np.nan):
    """State table in the reference layout: Array{Float64,3}(tot,d1,d2), iteration fastest."""
    return {k: np.full((tot,) + s, fill, dtype=np.float64, order="F") for k, s in table_shapes(V, R).items()}


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def uniform2(seed, it, site, elem, att=0):
    out = (C.c_double * 2)()
    lib().orc_uniform2(seed, it, site, elem, att, out)
    return out[0], out[1]


def normal(seed, it, site, elem, att=0):
    return lib().orc_normal(seed, it, site, elem, att)


def rhat(chains):
    """chains: (nsamp, nparams, nchains) -> (nparams,)   convergence.jl:4-65"""
    a = np.asfortranarray(chains, dtype=np.float64)
    out = np.empty(a.shape[1])
    lib().orc_rhat(a.ctypes.data_as(C.c_void_p), C.c_int(a.shape[0]), C.c_int(a.shape[1]), C.c_int(a.shape[2]),
                   out.ctypes.data_as(C.c_void_p))
    return out


class Oracle:
    """One chain of the CPU restatement operating on a reference-layout state table."""

    def __init__(self, X, y, R, tot, seed, chain=1, eta=1.01, zeta=1.0, iota=1.0, aDelta=1.0, bDelta=1.0,
                 nu=10, pdf_mode=0, cost_mode=0, table=None):
        self.X = np.asfortranarray(X, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        n, q = self.X.shape
        V = int(round((-1 + np.sqrt(1 + 8 * q)) / 2))
        assert V * (V + 1) // 2 == q
        self.n, self.q, self.V, self.R, self.tot = n, q, V, R, tot
        self.t = table if table is not None else new_table(tot, V, R)
        for k, s in table_shapes(V, R).items():
            a = self.t[k]
            assert a.shape == (tot,) + s and a.flags.f_contiguous and a.dtype == np.float64, k
        o = _Orc()
        o.n, o.V, o.R, o.q = n, V, R, q
        o.eta, o.zeta, o.iota, o.aDelta, o.bDelta, o.nu = eta, zeta, iota, aDelta, bDelta, float(nu)
        o.seed = int(seed) + int(chain)
        o.X = self.X.ctypes.data
        o.y = self.y.ctypes.data
        o.pdf_mode, o.cost_mode, o.tot, o.status = pdf_mode, cost_mode, tot, 0
        for k in COLUMNS:
            setattr(o, k, self.t[k].ctypes.data)
        o.iter = 1
        self.o = o
        self.p = C.byref(o)
        self.L = lib()

    # --- whole-chain entry points (1-based reference indices) -------------------------
    def init_prior(self):
        self.L.orc_init_prior(self.p)

    def run(self, first_index, nburn, total, purge_burn=None):
        r = self.L.orc_run(self.p, C.c_int(first_index), C.c_int(nburn), C.c_int(total),
                           C.c_int(purge_burn if purge_burn else 0))
        if r < 0:
            raise RuntimeError("oracle failed with status %d" % -r)
        return r

    def gibbs_sample(self, row, it):
        """row: 0-based row to write; it: global iteration id used for the RNG counter."""
        self.L.orc_gibbs_sample(self.p, C.c_int(row), C.c_uint32(it))

    def update(self, name, row, it):
        getattr(self.L, "orc_update_" + name)(self.p, C.c_int(row), C.c_uint32(it))

    # --- deterministic conditional parameters (for PIT pinning) -----------------------
    def tau2_params(self, row):
        out = (C.c_double * 2)()
        self.L.orc_tau2_params(self.p, C.c_int(row), out)
        return out[0], out[1]

    def theta_params(self, row):
        out = (C.c_double * 2)()
        self.L.orc_theta_params(self.p, C.c_int(row), out)
        return out[0], out[1]

    def mu_params(self, row):
        out = (C.c_double * 2)()
        self.L.orc_mu_params(self.p, C.c_int(row), out)
        return out[0], out[1]

    def node_params(self, row, k):
        R = self.R
        w = C.c_double()
        logit = C.c_double()
        mu_t = np.empty(R)
        Lc = np.empty((R, R), order="F")
        rc = self.L.orc_node_params(self.p, C.c_int(row), C.c_int(k), C.byref(w), mu_t.ctypes.data_as(C.c_void_p),
                                    Lc.ctypes.data_as(C.c_void_p), C.byref(logit))
        return rc, w.value, mu_t, Lc, logit.value

    def M_params(self, row):
        R = self.R
        Psi = np.empty((R, R), order="F")
        df = C.c_double()
        self.L.orc_M_params(self.p, C.c_int(row), Psi.ctypes.data_as(C.c_void_p), C.byref(df))
        return Psi, df.value

    def Lambda_params(self, row):
        probs = np.empty((self.R, 3))
        self.L.orc_Lambda_params(self.p, C.c_int(row), probs.ctypes.data_as(C.c_void_p))
        return probs

    def pi_alpha(self, row, r):
        out = (C.c_double * 3)()
        self.L.orc_pi_alpha(self.p, C.c_int(row), C.c_int(r), out)
        return np.array(out[:])

    def compute_W(self, row_u, row_lam):
        W = np.empty(self.q)
        self.L.orc_compute_W(self.p, C.c_int(row_u), C.c_int(row_lam), W.ctypes.data_as(C.c_void_p))
        return W

    def sample_gig(self, lam, chi, psi, it, elem):
        return self.L.orc_sample_gig(self.p, lam, chi, psi, it, elem)

    def gamma_draw(self, a, it, site, elem):
        return self.L.orc_gamma(self.p, a, it, site, elem)

    @property
    def status(self):
        return self.o.status

    @property
    def iter(self):
        return self.o.iter

    @iter.setter
    def iter(self, v):
        self.o.iter = v
