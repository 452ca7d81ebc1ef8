#!/usr/bin/env python3
"""Extract golden vectors from the reference's test data into small .npz fixtures.

Run in the build container only (needs /root/reference).  Nothing under tests/ or the
product reads /root/reference at test time; they read the .npz files written here.

Sources (all DATA files of the reference's own test-suite / examples, no source code):
  test/data/gen_test_results.jld2  -> golden_st.npz (st1,st2,st3), golden_res.npz, golden_res2.npz
       (file written by test/gen_tst_results.jl:238, read by test/init-tests.jl:39,
        toy-generate-samples-test.jl:16, test1-generate-samples-test.jl:8)
  test/data/test1.csv              -> test1_xy.npz      (test1-generate-samples-test.jl:3-5)
  examples/matrix_networks.csv + responses.csv + true_b.csv + true_xi.csv -> examples_xy.npz (BASELINE.json configs[0])

The JLD2 container is plain HDF5 (superblock v2 at byte 512, v2 object headers,
contiguous / compact layouts, no compression), so a minimal reader suffices; neither
Julia nor h5py exists in this image.  Layout facts: SURVEY.md Appendix A.
"""
import struct, sys, os
import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
COLS11 = ["tau2", "u", "xi", "gamma", "S", "theta", "Delta", "M", "mu", "lam", "pi"]
COLS14 = COLS11 + ["Sigma_inv", "invC", "mu_t"]


class JLD2:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        assert self.b[512:520] == b"\x89HDF\r\n\x1a\n", "no HDF5 superblock at 512"
        assert self.b[520] == 2, "superblock version"
        self.base, _ext, _eof, root = struct.unpack_from("<QQQQ", self.b, 512 + 12)
        self.root = root

    def messages(self, addr):
        """Yield (type, payload) for the v2 object header at file-relative addr."""
        p = addr + self.base
        b = self.b
        assert b[p:p + 4] == b"OHDR", (addr, b[p:p + 4])
        flags = b[p + 5]
        p += 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        szlen = 1 << (flags & 3)
        chunk = int.from_bytes(b[p:p + szlen], "little")
        p += szlen
        yield from self._chunk(p, p + chunk, flags)

    def _chunk(self, p, end, flags):
        b = self.b
        while p + 4 <= end:
            mtype = b[p]
            msize = struct.unpack_from("<H", b, p + 1)[0]
            p += 4
            if flags & 0x04:
                p += 2
            payload = b[p:p + msize]
            if mtype == 16:  # continuation: (offset, length) -> "OCHK" block
                off, ln = struct.unpack_from("<QQ", payload, 0)
                q = off + self.base
                assert b[q:q + 4] == b"OCHK"
                yield from self._chunk(q + 4, q + ln - 4, flags)
            else:
                yield mtype, payload
            p += msize

    def links(self, addr):
        out = {}
        for t, m in self.messages(addr):
            if t != 6:
                continue
            fl = m[1]
            p = 2
            ltype = 0
            if fl & 0x08:
                ltype = m[p]; p += 1
            if fl & 0x04:
                p += 8
            if fl & 0x10:
                p += 1
            nl = 1 << (fl & 3)
            n = int.from_bytes(m[p:p + nl], "little"); p += nl
            name = m[p:p + n].decode(); p += n
            if ltype == 0:
                out[name] = struct.unpack_from("<Q", m, p)[0]
        return out

    def dataset(self, addr):
        """Return (dims_as_stored, raw_bytes)."""
        dims, raw = (), None
        for t, m in self.messages(addr):
            if t == 1:
                ver, rank, fl = m[0], m[1], m[2]
                assert ver == 2
                dims = struct.unpack_from("<%dQ" % rank, m, 4)
            elif t == 8:
                assert m[0] == 4, "layout version"
                cls = m[1]
                if cls == 1:
                    a, sz = struct.unpack_from("<QQ", m, 2)
                    raw = self.b[a + self.base:a + self.base + sz]
                elif cls == 0:
                    sz = struct.unpack_from("<H", m, 2)[0]
                    raw = m[4:4 + sz]
                else:
                    raise ValueError("chunked layout not expected")
        return dims, raw

    def f64_array(self, addr):
        dims, raw = self.dataset(addr)
        a = np.frombuffer(raw, dtype="<f8")
        if dims:
            a = a.reshape(dims).transpose()  # -> Julia index order (iter, d1, d2)
        return np.array(a)

    def table(self, addr, names):
        _d, raw = self.dataset(addr)
        refs = struct.unpack("<%dQ" % (len(raw) // 8), raw)
        assert len(refs) == len(names), (len(refs), names)
        return {n: self.f64_array(r) for n, r in zip(names, refs)}

    def results(self, addr):
        _d, raw = self.dataset(addr)
        assert len(raw) == 40
        st, rx, rg, burn, samp = struct.unpack("<QQQqq", raw)
        out = self.table(st, COLS14)
        out["rhat_xi"] = self.table(rx, ["xi"])["xi"]
        out["rhat_gamma"] = self.table(rg, ["gamma"])["gamma"]
        out["burn_in"] = np.int64(burn)
        out["sampled"] = np.int64(samp)
        return out


def main():
    f = JLD2(os.path.join(REF, "test/data/gen_test_results.jld2"))
    root = f.links(f.root)
    print("root links:", sorted(root))

    st = {}
    for name in ("st1", "st2", "st3"):
        t = f.table(root[name], COLS11)
        for k, v in t.items():
            st[name + "_" + k] = v
    # parser self-check values recorded in SURVEY.md section 4
    assert abs(st["st1_theta"][0, 0, 0] - 0.5) < 1e-12
    assert abs(st["st1_u"][0, 0, 0] - 2.343829) < 1e-6
    assert abs(st["st1_M"][0, 0, 0] - 0.332580) < 1e-6
    assert abs(st["st1_gamma"][0, 0, 0] - 5.240301) < 1e-6
    assert abs(st["st2_tau2"][1, 0, 0] - 37.606403) < 1e-6
    assert abs(st["st3_tau2"][2, 0, 0] - 5.161772) < 1e-6
    assert abs(st["st3_mu"][2, 0, 0] - 12.225498) < 1e-6
    np.savez_compressed(os.path.join(OUT, "golden_st.npz"), **st)

    for name in ("res", "res2"):
        r = f.results(root[name])
        for dead in ("Sigma_inv", "invC", "mu_t"):   # never written by the reference (garbage)
            r.pop(dead)
        print(name, {k: getattr(v, "shape", v) for k, v in r.items()})
        np.savez_compressed(os.path.join(OUT, "golden_%s.npz" % name), **r)
        if name == "res":
            assert np.allclose(r["rhat_xi"], [1.134100, 1.004402, 1.051240, 0.993874], atol=1e-6)
            assert abs(r["tau2"].mean() - 2.6039790948) < 1e-9
        else:
            assert np.allclose(r["tau2"][:3, 0, 0], [1, 292.862539, 124.665318], atol=1e-6)
            assert abs(r["tau2"].mean() - 3.5876727047) < 1e-9
            assert abs(r["rhat_gamma"].max() - 1.064741) < 1e-6

    d = np.loadtxt(os.path.join(REF, "test/data/test1.csv"), delimiter=",", skiprows=1)
    assert d.shape == (70, 191)
    np.savez_compressed(os.path.join(OUT, "test1_xy.npz"), X=d[:, :190], y=d[:, 190])

    X = np.loadtxt(os.path.join(REF, "examples/matrix_networks.csv"), delimiter=",", skiprows=1)
    y = np.loadtxt(os.path.join(REF, "examples/responses.csv"), delimiter=",", skiprows=1)
    assert X.shape == (100, 466) and np.array_equal(X[:, 465], y)   # last csv column is y itself
    X = X[:, :465]                                                  # V=30 incl. diagonal
    # the truth the example data was simulated from (examples/true_b.csv: the 435 off-diagonal edge coefficients,
    # examples/true_xi.csv: which of the 30 nodes are influential)
    true_b = np.loadtxt(os.path.join(REF, "examples/true_b.csv"), skiprows=1)
    true_xi = np.array([ln.strip() == "true" for ln in open(os.path.join(REF, "examples/true_xi.csv")).read().split()[1:]], dtype=float)
    assert true_b.shape == (435,) and true_xi.shape == (30,)
    print("examples", X.shape, y.shape, true_b.shape, true_xi.sum())
    np.savez_compressed(os.path.join(OUT, "examples_xy.npz"), X=X, y=y, true_b=true_b, true_xi=true_xi)
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
