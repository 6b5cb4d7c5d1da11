"""ctypes bindings of the TEST-ONLY oracle (oracle/libasm_oracle.so) and, where present, of the real reference
build (oracle/_ref/libasm_ref.so).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libasm_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libasm_ref.so")
REF_SIMD_SO = os.path.join(ROOT, "oracle", "_ref", "libasm_ref_simd.so")
# state the reference harness pins before a batch (its warm-up pair): final_ED, lane distance, converge_ED
SIMD_WARM_STATE = (1, 1, 2)
CIGAR_STRIDE = 768
DEFAULT_PROBS = (0.80, 0.20 / 3, 0.40 / 3)

_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def _batch_args(hb):
    reads = np.ascontiguousarray(hb.reads, np.uint8)
    refs = np.ascontiguousarray(hb.refs, np.uint8)
    ro = np.ascontiguousarray(hb.read_off, np.uint32)
    fo = np.ascontiguousarray(hb.ref_off, np.uint32)
    keep = (reads, refs, ro, fo)
    return keep, (hb.n, reads.ctypes.data, ro.ctypes.data, refs.ctypes.data, fo.ctypes.data)


def _cigars(buf, n, stride):
    raw = buf.tobytes()
    return [raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() for i in range(n)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        b5 = [_i64, _vp, _vp, _vp, _vp]
        lib.orc_greedy_batch.argtypes = b5 + [_i] * 4 + [_vp, _i, _vp, _vp, _i, _vp]
        lib.orc_greedy_batch_typed.argtypes = b5 + [_i] * 4 + [_vp, _i, _i, _vp, _vp, _i, _vp]
        lib.orc_greedy_views.argtypes = b5 + [_i, _vp]
        lib.orc_leap_batch.argtypes = b5 + [_i] * 4 + [_vp]
        lib.orc_nw_batch.argtypes = b5 + [_i] * 3 + [_vp]
        lib.orc_levenshtein_batch.argtypes = b5 + [_vp]
        lib.orc_nw_cigar_batch.argtypes = b5 + [_i] * 3 + [_vp, _vp, _i]
        lib.orc_coverage_batch.argtypes = b5 + [_vp, _i, _i, _vp, _i, _i, _vp]
        lib.orc_set_threads.argtypes = [_i]
        lib.orc_simd_ed_batch.argtypes = b5 + [_i] * 3 + [_vp] * 4
        lib.orc_shd_batch.argtypes = b5 + [_i, _vp]

    def set_threads(self, n):
        return self.lib.orc_set_threads(n)

    def greedy(self, hb, k=3, x=1, o=1, e=1, probs=DEFAULT_PROBS, mode=1, cigars=False, steps=False, semi=False):
        keep, args = _batch_args(hb)
        costs = np.zeros(hb.n, np.int32)
        pr = np.array(probs, np.float64)
        cg = np.zeros(hb.n * CIGAR_STRIDE + 1, np.uint8) if cigars else None
        st = np.zeros(hb.n, np.int32) if steps else None
        rc = self.lib.orc_greedy_batch_typed(*args, k, x, o, e, pr.ctypes.data, mode, 1 if semi else 0, costs.ctypes.data,
                                             cg.ctypes.data if cigars else None, CIGAR_STRIDE,
                                             st.ctypes.data if steps else None)
        assert rc == 0, rc
        out = [costs]
        if cigars:
            out.append(_cigars(cg, hb.n, CIGAR_STRIDE))
        if steps:
            out.append(st)
        return out[0] if len(out) == 1 else tuple(out)

    def greedy_views(self, hb, mode):
        keep, args = _batch_args(hb)
        views = np.zeros(hb.n * 256 + 1, np.uint8)
        assert self.lib.orc_greedy_views(*args, mode, views.ctypes.data) == 0
        return views[:hb.n * 256].reshape(hb.n, 2, 128)

    def leap(self, hb, k=3, x=1, o=1, e=1):
        keep, args = _batch_args(hb)
        eds = np.zeros(hb.n, np.int32)
        rc = self.lib.orc_leap_batch(*args, k, x, o, e, eds.ctypes.data)
        assert rc == 0, rc
        return eds

    def nw(self, hb, x=1, o=1, e=1):
        keep, args = _batch_args(hb)
        pen = np.zeros(hb.n, np.int32)
        assert self.lib.orc_nw_batch(*args, x, o, e, pen.ctypes.data) == 0
        return pen

    def levenshtein(self, hb):
        keep, args = _batch_args(hb)
        d = np.zeros(hb.n, np.int32)
        assert self.lib.orc_levenshtein_batch(*args, d.ctypes.data) == 0
        return d

    def nw_cigar(self, hb, x=1, o=1, e=1):
        keep, args = _batch_args(hb)
        pen = np.zeros(hb.n, np.int32)
        cg = np.zeros(hb.n * CIGAR_STRIDE + 1, np.uint8)
        assert self.lib.orc_nw_cigar_batch(*args, x, o, e, pen.ctypes.data, cg.ctypes.data, CIGAR_STRIDE) == 0
        return pen, _cigars(cg, hb.n, CIGAR_STRIDE)

    def simd_ed(self, hb, ed_t=3, shd=True, mode=0, state=SIMD_WARM_STATE):
        """(ed, ed_raw, pass): ed = converge_ED when the pair passes else -1; ed_raw = get_ED() whatever the verdict."""
        keep, args = _batch_args(hb)
        ed = np.zeros(hb.n, np.int32)
        raw = np.zeros(hb.n, np.int32)
        ps = np.zeros(hb.n, np.uint8)
        st = np.array(state, np.int32)
        rc = self.lib.orc_simd_ed_batch(*args, ed_t, 1 if shd else 0, mode, st.ctypes.data, ed.ctypes.data,
                                        raw.ctypes.data, ps.ctypes.data)
        assert rc == 0, rc
        return ed, raw, ps

    def shd(self, hb, max_error=3):
        keep, args = _batch_args(hb)
        ps = np.zeros(hb.n, np.int32)
        assert self.lib.orc_shd_batch(*args, max_error, ps.ctypes.data) == 0
        return ps

    def coverage(self, hb, cigars1, thr1, cigars2, thr2):
        keep, args = _batch_args(hb)

        def pack(cs):
            buf = np.zeros(hb.n * CIGAR_STRIDE + 1, np.uint8)
            for i, c in enumerate(cs):
                b = c.encode()
                buf[i * CIGAR_STRIDE:i * CIGAR_STRIDE + len(b)] = np.frombuffer(b, np.uint8)
            return buf

        b1, b2 = pack(cigars1), pack(cigars2)
        out = np.zeros(hb.n, np.uint8)
        assert self.lib.orc_coverage_batch(*args, b1.ctypes.data, CIGAR_STRIDE, thr1, b2.ctypes.data, CIGAR_STRIDE,
                                           thr2, out.ctypes.data) == 0
        return out


class ReferenceSimd:
    """The real SIMD_ED / SHD sources compiled in place — this container only (the built .so travels to the GPU box)."""

    def __init__(self, lib):
        self.lib = lib
        b5 = [_i64, _vp, _vp, _vp, _vp]
        lib.ref_simd_ed_batch.argtypes = b5 + [_i, _i, _vp, _vp]
        lib.ref_shd_batch.argtypes = b5 + [_i, _vp]

    def simd_ed(self, hb, ed_t=3, shd=True):
        """(get_ED() raw, check_pass()) per pair, run in batch order after the harness's warm-up pair."""
        keep, args = _batch_args(hb)
        ed = np.zeros(hb.n, np.int32)
        ps = np.zeros(hb.n, np.uint8)
        assert self.lib.ref_simd_ed_batch(*args, ed_t, 1 if shd else 0, ed.ctypes.data, ps.ctypes.data) == 0
        return ed, ps

    def shd(self, hb, max_error=3):
        keep, args = _batch_args(hb)
        ps = np.zeros(hb.n, np.int32)
        assert self.lib.ref_shd_batch(*args, max_error, ps.ctypes.data) == 0
        return ps


class Reference:
    """The real reference (Greedy + LEAP) compiled in place from /root/reference — this container only."""

    def __init__(self, lib):
        self.lib = lib
        b5 = [_i64, _vp, _vp, _vp, _vp]
        lib.ref_greedy_batch.argtypes = b5 + [_i] * 4 + [ctypes.c_double] * 3 + [_i, _vp, _vp, _i, _vp]
        lib.ref_greedy_batch_typed.argtypes = b5 + [_i] * 4 + [ctypes.c_double] * 3 + [_i, _i, _vp, _vp, _i, _vp]
        lib.ref_leap_batch.argtypes = b5 + [_i] * 4 + [_vp, _vp]
        lib.ref_leap_batch_ex.argtypes = b5 + [_i] * 4 + [_vp, _vp, _i]
        lib.ref_convert2bit1.argtypes = [_vp, _vp, _vp]

    def greedy(self, hb, k=3, x=1, o=1, e=1, probs=DEFAULT_PROBS, mode=1, cigars=False, views=False, semi=False):
        keep, args = _batch_args(hb)
        costs = np.zeros(hb.n, np.int32)
        cg = np.zeros(hb.n * CIGAR_STRIDE + 1, np.uint8) if cigars else None
        vw = np.zeros(hb.n * 256 + 1, np.uint8) if views else None
        rc = self.lib.ref_greedy_batch_typed(*args, k, x, o, e, *probs, mode, 1 if semi else 0, costs.ctypes.data,
                                             cg.ctypes.data if cigars else None, CIGAR_STRIDE,
                                             vw.ctypes.data if views else None)
        assert rc == 0
        out = [costs]
        if cigars:
            out.append(_cigars(cg, hb.n, CIGAR_STRIDE))
        if views:
            out.append(vw[:hb.n * 256].reshape(hb.n, 2, 128))
        return out[0] if len(out) == 1 else tuple(out)

    def leap(self, hb, k=3, x=1, o=1, e=1, full=False):
        """full=True also runs backtrack() + get_CIGAR(), as the harness's timed LEAP region does."""
        keep, args = _batch_args(hb)
        eds = np.zeros(hb.n, np.int32)
        assert self.lib.ref_leap_batch_ex(*args, k, x, o, e, eds.ctypes.data, None, 1 if full else 0) == 0
        return eds

    def convert2bit1(self, buf128):
        buf = np.array(buf128, np.uint8).copy()
        b0 = np.zeros(16, np.uint8)
        b1 = np.zeros(16, np.uint8)
        self.lib.ref_convert2bit1(buf.ctypes.data, b0.ctypes.data, b1.ctypes.data)
        return buf, b0, b1


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    return Oracle(ctypes.CDLL(ORACLE_SO))


def have_reference():
    return os.path.exists(REF_SO)


def load_reference():
    return Reference(ctypes.CDLL(REF_SO))


def have_reference_simd():
    return os.path.exists(REF_SIMD_SO)


def load_reference_simd():
    return ReferenceSimd(ctypes.CDLL(REF_SIMD_SO))
