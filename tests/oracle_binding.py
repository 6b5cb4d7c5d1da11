"""ctypes bindings of the TEST-ONLY oracle (oracle/libasm_oracle.so) and, where present, of the real reference
build (oracle/_ref/libasm_ref.so).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ASM_ORACLE_LIB: another build of the same checker (tests/test_sanitizers.py points it at the ASan + UBSan build)
ORACLE_SO = os.environ.get("ASM_ORACLE_LIB") or os.path.join(ROOT, "oracle", "libasm_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libasm_ref.so")
REF_SIMD_SO = os.path.join(ROOT, "oracle", "_ref", "libasm_ref_simd.so")
REF_DATASET = os.path.join(ROOT, "oracle", "_ref", "ref_dataset")  # the reference's own Dataset generator with a settable seed
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


def base_code(c):
    """bit_convert.cpp:340-355: A=00 C=01 G=10 T=11, anything else 00."""
    return {ord("C"): 1, ord("G"): 2, ord("T"): 3}.get(int(c), 0)


_PERM_P = (0, 2, 1, 3, 4, 6, 5, 7)


def tail_slot_after(slot, n):
    """Where the byte sitting in `slot` before a pair's conversion sits n pairs later: after[q] = before[8*(q%16) + P[q//16]]
    (bit_convert.cpp:265-330), so a byte moves from y to the q with SRC[q] = y; the permutation has order 10."""
    inv = {8 * (q % 16) + _PERM_P[q // 16]: q for q in range(128)}
    for _ in range(n % 10):
        slot = inv[slot]
    return slot


def codes_to_buffers(state):
    """2-bit codes -> bytes the reference's conversion maps to them (0 -> NUL, as a fresh buffer)."""
    lut = np.array([0, ord("C"), ord("G"), ord("T")], np.uint8)
    return lut[np.asarray(state, np.uint8)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        b5 = [_i64, _vp, _vp, _vp, _vp]
        lib.orc_greedy_batch.argtypes = b5 + [_i] * 4 + [_vp, _i, _vp, _vp, _i, _vp]
        lib.orc_greedy_batch_typed.argtypes = b5 + [_i] * 4 + [_vp, _i, _i, _vp, _vp, _i, _vp]
        lib.orc_greedy_views.argtypes = b5 + [_i, _vp]
        lib.orc_leap_batch.argtypes = b5 + [_i] * 4 + [_vp]
        lib.orc_leap_mode_batch.argtypes = b5 + [_i] * 5 + [_vp]
        lib.orc_nw_batch.argtypes = b5 + [_i] * 3 + [_vp]
        lib.orc_levenshtein_batch.argtypes = b5 + [_vp]
        lib.orc_nw_cigar_batch.argtypes = b5 + [_i] * 3 + [_vp, _vp, _i]
        lib.orc_coverage_batch.argtypes = b5 + [_vp, _i, _i, _vp, _i, _i, _vp]
        lib.orc_set_threads.argtypes = [_i]
        lib.orc_simd_ed_batch.argtypes = b5 + [_i] * 3 + [_vp] * 4
        lib.orc_shd_batch.argtypes = b5 + [_i, _vp]
        lib.orc_simd_ed_edmode_batch.argtypes = b5 + [_i] * 4 + [_vp] * 4
        lib.orc_simd_ed_affine_batch.argtypes = b5 + [_i] * 5 + [_vp, _vp]
        lib.orc_simd_ed_affine_shd_batch.argtypes = b5 + [_i] * 7 + [_vp, _vp]
        lib.orc_simd_ed_affine_mode_batch.argtypes = b5 + [_i] * 8 + [_vp, _vp]

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

    # ---- sequential mode in pieces (shards of a file / chunks of a stream) ----
    def set_initial_buffers(self, ab=None):
        """Content of the reference's A and B buffers (uint8[256]) before the next greedy call's first pair; None = zeros."""
        self.lib.orc_greedy_set_initial_buffers.argtypes = [_vp]
        if ab is None:
            self.lib.orc_greedy_set_initial_buffers(None)
        else:
            buf = np.ascontiguousarray(ab, np.uint8)
            assert buf.size == 256
            self.lib.orc_greedy_set_initial_buffers(buf.ctypes.data)

    def final_buffers(self):
        self.lib.orc_greedy_get_final_buffers.argtypes = [_vp]
        out = np.zeros(256, np.uint8)
        self.lib.orc_greedy_get_final_buffers(out.ctypes.data)
        return out

    def tail_summary(self, hb):
        """What the batch does to the buffers, in the product's terms (asm_batch_tail_summary): summary[side*128 + s] = code of
        the last character written on the trajectory that starts in slot s, 0xFF when the batch never writes on it.  Found by
        running the buffer model from sentinel bytes and seeing which survive."""
        self.set_initial_buffers(np.full(256, 0xEE, np.uint8))
        try:
            self.greedy_views(hb, 0)
            fin = self.final_buffers()
        finally:
            self.set_initial_buffers(None)
        out = np.full(256, 0xFF, np.uint8)
        for side in range(2):
            for s in range(128):
                c = fin[side * 128 + tail_slot_after(s, hb.n)]
                if c != 0xEE:
                    out[side * 128 + s] = base_code(c)
        return out

    def reference_dataset(self, n, length, err, seed, mismatch_rate=0.96, exact=True):
        """Pairs drawn the reference's way (oracle/asm_oracle_dataset.c: Dataset over glibc's rand() after srand(seed)) as a
        batch in the C ABI's layout (arrays are what approximate_string_matching_amd.HostBatch holds)."""
        import math

        cap = n * (length + int(math.ceil(length * float(np.float32(err)))) + 2) + 16
        reads = np.zeros(n * length + 16, np.uint8)
        refs = np.zeros(cap, np.uint8)
        ro = np.zeros(n + 1, np.uint32)
        fo = np.zeros(n + 1, np.uint32)
        self.lib.orc_reference_dataset_ex.argtypes = [_i64, _i, ctypes.c_float, ctypes.c_float, _i, ctypes.c_uint, _vp, _vp, _vp, _vp]
        rc = self.lib.orc_reference_dataset_ex(n, length, err, mismatch_rate, 1 if exact else 0, seed, reads.ctypes.data, ro.ctypes.data,
                                               refs.ctypes.data, fo.ctypes.data)
        assert rc == 0, rc
        return reads[:int(ro[-1])], ro, refs[:int(fo[-1])], fo

    def glibc_rand_stream(self, seed, count):
        out = np.zeros(count, np.int32)
        self.lib.orc_glibc_rand_stream.argtypes = [ctypes.c_uint, _i, _vp]
        self.lib.orc_glibc_rand_stream(seed, count, out.ctypes.data)
        return out

    def greedy_views(self, hb, mode):
        keep, args = _batch_args(hb)
        views = np.zeros(hb.n * 256 + 1, np.uint8)
        assert self.lib.orc_greedy_views(*args, mode, views.ctypes.data) == 0
        return views[:hb.n * 256].reshape(hb.n, 2, 128)

    def leap(self, hb, k=3, x=1, o=1, e=1, mode=0):
        """mode: LV::init's ED_modes in the oracle's numbering — 0 GLOBAL (the harness), 1 LOCAL, 2 SEMI_FREE_BEGIN, 3 SEMI_FREE_END."""
        keep, args = _batch_args(hb)
        eds = np.zeros(hb.n, np.int32)
        rc = self.lib.orc_leap_mode_batch(*args, k, x, o, e, mode, eds.ctypes.data)
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

    def simd_ed(self, hb, ed_t=3, shd=True, mode=0, state=SIMD_WARM_STATE, ed_mode=0):
        """(ed, ed_raw, pass): ed = get_ED() when the pair passes else -1; ed_raw = get_ED() whatever the verdict.  ed_mode:
        init_levenshtein's ED_modes in the oracle's numbering (0 GLOBAL, 1 LOCAL, 2 SEMI_FREE_BEGIN, 3 SEMI_FREE_END)."""
        keep, args = _batch_args(hb)
        ed = np.zeros(hb.n, np.int32)
        raw = np.zeros(hb.n, np.int32)
        ps = np.zeros(hb.n, np.uint8)
        st = np.array(state, np.int32)
        rc = self.lib.orc_simd_ed_edmode_batch(*args, ed_t, 1 if shd else 0, mode, int(ed_mode), st.ctypes.data, ed.ctypes.data,
                                               raw.ctypes.data, ps.ctypes.data)
        assert rc == 0, rc
        return ed, raw, ps

    def simd_ed_affine(self, hb, gap_t=3, af_t=60, x=2, o=3, e=1, shd_t=None, mode=0):
        """(ed, pass): SIMD_ED affine mode, clean; ed = get_ED() (1000000 for a pair exact at e = 0) when the pair passes else -1.
        shd_t: init_affine's SHD_threshold with SHD_enable = true (None: SHD off, init_affine's default); mode: its ED_modes in
        the oracle's numbering (0 GLOBAL, 1 LOCAL, 2 SEMI_FREE_BEGIN, 3 SEMI_FREE_END)."""
        keep, args = _batch_args(hb)
        ed = np.zeros(hb.n, np.int32)
        ps = np.zeros(hb.n, np.uint8)
        rc = self.lib.orc_simd_ed_affine_mode_batch(*args, gap_t, af_t, x, o, e, 0 if shd_t is None else 1,
                                                    0 if shd_t is None else int(shd_t), int(mode), ed.ctypes.data, ps.ctypes.data)
        assert rc == 0, rc
        return ed, ps

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
        lib.ref_simd_ed_edmode_batch.argtypes = b5 + [_i] * 3 + [ctypes.c_char_p, ctypes.c_char_p, _vp, _vp]
        lib.ref_simd_ed_affine_batch.argtypes = b5 + [_i] * 5 + [_vp, _vp]
        lib.ref_simd_ed_affine_shd_batch.argtypes = b5 + [_i] * 7 + [_vp, _vp]
        lib.ref_simd_ed_affine_mode_batch.argtypes = b5 + [_i] * 8 + [_vp, _vp]

    def simd_ed(self, hb, ed_t=3, shd=True):
        """(get_ED() raw, check_pass()) per pair, run in batch order after the harness's warm-up pair."""
        keep, args = _batch_args(hb)
        ed = np.zeros(hb.n, np.int32)
        ps = np.zeros(hb.n, np.uint8)
        assert self.lib.ref_simd_ed_batch(*args, ed_t, 1 if shd else 0, ed.ctypes.data, ps.ctypes.data) == 0
        return ed, ps

    def simd_ed_edmode(self, hb, ed_t, shd, ed_mode, warm=("ACGT", "ACTT")):
        """(get_ED(), check_pass()) as run, one object, init_levenshtein(ed_t, ED_modes ed_mode, shd), after the warm-up pair."""
        keep, args = _batch_args(hb)
        ed = np.zeros(hb.n, np.int32)
        ps = np.zeros(hb.n, np.uint8)
        assert self.lib.ref_simd_ed_edmode_batch(*args, ed_t, 1 if shd else 0, int(ed_mode), warm[0].encode(), warm[1].encode(),
                                                 ed.ctypes.data, ps.ctypes.data) == 0
        return ed, ps

    def simd_ed_affine(self, hb, gap_t=3, af_t=60, x=2, o=3, e=1, shd_t=None, mode=0):
        """(get_ED(), check_pass()) per pair with init_affine before every pair (clean tables); shd_t: SHD_enable = true with that
        SHD_threshold; mode: ED_modes in the oracle's numbering."""
        keep, args = _batch_args(hb)
        ed = np.zeros(hb.n, np.int32)
        ps = np.zeros(hb.n, np.uint8)
        assert self.lib.ref_simd_ed_affine_mode_batch(*args, gap_t, af_t, x, o, e, 0 if shd_t is None else 1,
                                                      0 if shd_t is None else int(shd_t), int(mode), ed.ctypes.data, ps.ctypes.data) == 0
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
        lib.ref_leap_mode_batch.argtypes = b5 + [_i] * 6 + [_vp]
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

    def leap_mode(self, hb, k=3, x=1, o=1, e=1, mode=0, clean=True):
        """LV with ED_modes `mode` (oracle numbering); clean: init() before every pair, else one object with reset() as the harness."""
        keep, args = _batch_args(hb)
        eds = np.zeros(hb.n, np.int32)
        assert self.lib.ref_leap_mode_batch(*args, k, x, o, e, mode, 1 if clean else 0, eds.ctypes.data) == 0
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
