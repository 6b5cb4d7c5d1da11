"""MI355X-native batched pair aligner — Python host side over the C ABI (include/asm_mi355x.h).

This package is plumbing: it binds libasm_mi355x.so (HIP kernels + C ABI, built in-tree by
`make -C approximate-string-matching_amd lib`) with ctypes and mirrors the reference's per-pair interface
for the ONE path this repo accelerates — `benchmark::_run_benchmark`
(/root/reference/GASMA/benchmark/benchmark_utils.h:231-259: NW, LEAP, Greedy per read pair) — as whole-batch
calls.  There is no CPU fallback: every compute call needs the library and a HIP device and raises otherwise.

Import name: the directory is `approximate-string-matching_amd`; `import approximate_string_matching_amd`
works through the loader module of that name at the repo root.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASM_MI355X_LIB") or os.path.join(_HERE, "libasm_mi355x.so")  # override: A/B builds of the same ABI
HEADER_PATH = os.path.join(_HERE, "..", "include", "asm_mi355x.h")

NW, LEAP, GREEDY = 0, 1, 2
ALIGNER_NAMES = {NW: "nw", LEAP: "leap", GREEDY: "greedy"}
GREEDY_SEQUENTIAL, GREEDY_CLEAN = 0, 1
FILTER_SEQUENTIAL, FILTER_CLEAN = 0, 1
ALIGN_GLOBAL, ALIGN_SEMI_GLOBAL = 0, 1
LEAP_GLOBAL, LEAP_LOCAL, LEAP_SEMI_FREE_BEGIN, LEAP_SEMI_FREE_END = 0, 1, 2, 3  # asm_params.leap_mode: LV::init's ED_modes
GEN_EXACT_ERRORS, GEN_PER_BASE, GEN_UP_TO_ERRORS = 0, 1, 2
GREEDY_MAX_LENGTH = 128
LEAP_MAX_LENGTH = 256
MAX_LENGTH = 512


class AsmError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"asm_mi355x error {code}: {message}")
        self.code = code


class Params(ctypes.Structure):
    """asm_params: constructor arguments of `benchmark` (benchmark_utils.h:263-289) and of hurdle_matrix
    (hurdle_matrix.h:552-559)."""

    _fields_ = [("k", ctypes.c_int32), ("x", ctypes.c_int32), ("o", ctypes.c_int32), ("e", ctypes.c_int32),
                ("p_match", ctypes.c_double), ("p_mismatch", ctypes.c_double), ("p_indel", ctypes.c_double),
                ("alignment_type", ctypes.c_int32), ("leap_mode", ctypes.c_int32)]

    @classmethod
    def default(cls, k: int = 3, x: int = 1, o: int = 1, e: int = 1, p_match: float = 0.80,
                p_mismatch: float = 0.20 / 3, p_indel: float = 0.40 / 3, alignment_type: int = 0, leap_mode: int = 0) -> "Params":
        """alignment_type: ALIGN_GLOBAL (0) or ALIGN_SEMI_GLOBAL (1) — hurdle_matrix's alignment_type_t; Greedy only.
        leap_mode: LEAP_GLOBAL (0, the harness) / LEAP_LOCAL / LEAP_SEMI_FREE_BEGIN / LEAP_SEMI_FREE_END — LV::init's ED_modes."""
        return cls(k, x, o, e, p_match, p_mismatch, p_indel, alignment_type, leap_mode)


class StreamStats(ctypes.Structure):
    """asm_stream_stats: what asm_stream_seq_file did."""

    _fields_ = [("pairs", ctypes.c_int64), ("chunks", ctypes.c_int64), ("bytes", ctypes.c_int64),
                ("counters", ctypes.c_ulonglong * 4), ("seconds", ctypes.c_double), ("seconds_read", ctypes.c_double),
                ("max_length", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


class GenConfig(ctypes.Structure):
    """asm_gen_config: seeded restatement of `Dataset` (benchmark_dataset.h:61-253)."""

    _fields_ = [("seed", ctypes.c_uint64), ("kind", ctypes.c_int32), ("len_lo", ctypes.c_int32),
                ("len_hi", ctypes.c_int32), ("err", ctypes.c_float), ("mismatch_rate", ctypes.c_float),
                ("p_sub", ctypes.c_float), ("p_ins", ctypes.c_float), ("p_del", ctypes.c_float)]

    @classmethod
    def exact(cls, seed: int, length: int, err: float, mismatch_rate: float = 0.96, length_hi: Optional[int] = None):
        """Dataset(num_reads, length, err, 0.96, exact=true) — benchmark.cpp:19."""
        return cls(seed, GEN_EXACT_ERRORS, length, length_hi if length_hi is not None else length, err,
                   mismatch_rate, 0.0, 0.0, 0.0)

    @classmethod
    def up_to(cls, seed: int, length: int, err: float, mismatch_rate: float = 0.96, length_hi: Optional[int] = None):
        """Dataset(..., exact=false), the "lt_eq" files of GASMA/benchmark/README.md: 0 .. ceil(L*err) - 1 edit operations,
        uniformly (benchmark_dataset.h:153-156)."""
        return cls(seed, GEN_UP_TO_ERRORS, length, length_hi if length_hi is not None else length, err,
                   mismatch_rate, 0.0, 0.0, 0.0)

    @classmethod
    def per_base(cls, seed: int, length: int, p_sub: float, p_ins: float, p_del: float,
                 length_hi: Optional[int] = None):
        """Independent per-base events; SRR611076-shaped rates are README.md:73-76 of the reference."""
        return cls(seed, GEN_PER_BASE, length, length_hi if length_hi is not None else length, 0.0, 0.0, p_sub,
                   p_ins, p_del)


# The named workloads of BASELINE.json `configs`.
def workload(name: str) -> Tuple[GenConfig, int, Params]:
    """-> (generator config, number of pairs, aligner params) for C1..C5."""
    if name == "C1":
        return GenConfig.exact(1, 100, 0.05), 10_000, Params.default(k=3)
    if name == "C2":
        return GenConfig.exact(2, 100, 0.10), 1_000_000, Params.default(k=3)
    if name == "C3":
        return GenConfig.exact(3, 150, 0.20), 10_000_000, Params.default(k=30)
    if name == "C4":
        return GenConfig.per_base(4, 100, 0.02452, 0.000468, 0.000553), 10_000_000, Params.default(k=3)
    if name == "C5":
        return GenConfig.exact(5, 64, 0.10, length_hi=300), 10_000_000, Params.default(k=3)
    raise KeyError(name)


_lib = None


def load_library() -> ctypes.CDLL:
    """Loads the in-tree C-ABI library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsmError(-2, f"{LIB_PATH} is missing — build it with `make -C {_HERE} lib` "
                           "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32, i64, u64p = c.c_void_p, c.c_int, c.c_int64, c.POINTER(c.c_ulonglong)
    sigs = {
        "asm_version": (c.c_char_p, []),
        "asm_default_params": (None, [c.POINTER(Params)]),
        "asm_device_count": (i32, []),
        "asm_create": (i32, [c.POINTER(vp), i32]),
        "asm_destroy": (i32, [vp]),
        "asm_last_error": (c.c_char_p, [vp]),
        "asm_set_stream": (i32, [vp, vp]),
        "asm_reset_stream": (i32, [vp]),
        "asm_synchronize": (i32, [vp]),
        "asm_generate_pairs": (i32, [c.POINTER(GenConfig), i64, i64, vp, vp, vp, c.c_size_t, vp, c.c_size_t]),
        "asm_batch_upload": (i32, [vp, i64, vp, vp, vp, vp, i32, c.POINTER(vp)]),
        "asm_batch_generate": (i32, [vp, c.POINTER(GenConfig), i64, i64, i32, c.POINTER(vp)]),
        "asm_reference_upload": (i32, [vp, vp, c.c_size_t, c.POINTER(vp)]),
        "asm_reference_free": (i32, [vp, vp]),
        "asm_batch_from_hits": (i32, [vp, vp, i64, vp, vp, vp, i32, c.POINTER(vp)]),
        "asm_batch_free": (i32, [vp, vp]),
        "asm_batch_from_text": (i32, [vp, vp, c.c_size_t, i32, c.POINTER(vp)]),
        "asm_stream_seq_file": (i32, [vp, c.c_char_p, c.POINTER(Params), i32, i32, i64, i64, vp, vp, vp, i64, vp, i64,
                                      c.POINTER(StreamStats)]),
        "asm_batch_tail_summary": (i32, [vp, vp, vp]),
        "asm_tail_state_advance": (i32, [vp, vp, i64]),
        "asm_batch_resolve_tails": (i32, [vp, vp, vp]),
        "asm_batch_size": (i64, [vp]),
        "asm_batch_max_length": (i32, [vp]),
        "asm_batch_text_bytes": (i64, [vp]),
        "asm_batch_download": (i32, [vp, vp, vp, vp, vp, c.c_size_t, vp, c.c_size_t]),
        "asm_batch_pack_async": (i32, [vp, vp]),
        "asm_align_batch_async": (i32, [vp, vp, i32, c.POINTER(Params), vp]),
        "asm_align_batch_hinted_async": (i32, [vp, vp, i32, c.POINTER(Params), vp, vp]),
        "asm_align_batch": (i32, [vp, i32, i64, vp, vp, vp, vp, c.POINTER(Params), i32, vp]),
        "asm_greedy_cigar_batch_async": (i32, [vp, vp, c.POINTER(Params), vp, vp, i32, vp]),
        "asm_cigar_format": (i32, [vp, i32, i32, vp, c.c_size_t]),
        "asm_coverage": (i32, [vp, vp, c.POINTER(Params), vp, i32, vp, i32, vp, vp, i32, vp, vp]),
        "asm_simd_ed_batch_async": (i32, [vp, vp, i32, i32, i32, vp, vp]),
        "asm_simd_ed_mode_batch_async": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
        "asm_simd_ed_affine_batch_async": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
        "asm_simd_ed_affine_shd_batch_async": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        "asm_simd_ed_affine_mode_batch_async": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
        "asm_shd_filter_batch_async": (i32, [vp, vp, i32, vp]),
        "asm_pipeline_join_async": (i32, [vp]),
        "asm_profile_enable": (i32, [vp, i32, c.c_uint32]),
        "asm_profile_read": (i32, [vp, vp, i32, vp]),
        "asm_count_equal_async": (i32, [vp, vp, vp, i64, vp]),
        "asm_accuracy_async": (i32, [vp, vp, vp, vp, vp, i64, vp]),
        "asm_run_benchmark_async": (i32, [vp, vp, c.POINTER(Params), i32, vp, vp, vp, vp, vp]),
        "asm_device_malloc": (i32, [vp, c.c_size_t, c.POINTER(vp)]),
        "asm_device_free": (i32, [vp, vp]),
        "asm_memcpy_d2h": (i32, [vp, vp, vp, c.c_size_t]),
        "asm_memcpy_h2d": (i32, [vp, vp, vp, c.c_size_t]),
        "asm_memset_async": (i32, [vp, vp, i32, c.c_size_t]),
        "asm_timer_create": (i32, [vp, c.POINTER(vp)]),
        "asm_timer_start": (i32, [vp, vp]),
        "asm_timer_stop": (i32, [vp, vp]),
        "asm_timer_elapsed_ms": (i32, [vp, vp, c.POINTER(c.c_float)]),
        "asm_timer_destroy": (i32, [vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    lib._asm_symbols = tuple(sigs)
    _lib = lib
    return lib


def device_count() -> int:
    return int(load_library().asm_device_count())


@dataclass
class HostBatch:
    """A batch of read pairs in the C ABI's layout: concatenated ASCII + n+1 prefix offsets."""

    reads: np.ndarray     # uint8
    read_off: np.ndarray  # uint32, n+1
    refs: np.ndarray      # uint8
    ref_off: np.ndarray   # uint32, n+1

    @property
    def n(self) -> int:
        return int(self.read_off.shape[0] - 1)

    @classmethod
    def from_strings(cls, pairs: Iterable[Tuple[str, str]]) -> "HostBatch":
        pairs = list(pairs)
        ro = np.zeros(len(pairs) + 1, np.uint32)
        fo = np.zeros(len(pairs) + 1, np.uint32)
        if pairs:
            ro[1:] = np.cumsum([len(p[0]) for p in pairs])
            fo[1:] = np.cumsum([len(p[1]) for p in pairs])
        reads = np.frombuffer("".join(p[0] for p in pairs).encode("ascii"), np.uint8).copy()
        refs = np.frombuffer("".join(p[1] for p in pairs).encode("ascii"), np.uint8).copy()
        return cls(reads, ro, refs, fo)

    def pair(self, i: int) -> Tuple[str, str]:
        a = self.reads[self.read_off[i]:self.read_off[i + 1]].tobytes().decode("ascii")
        b = self.refs[self.ref_off[i]:self.ref_off[i + 1]].tobytes().decode("ascii")
        return a, b

    def slice(self, lo: int, hi: int) -> "HostBatch":
        ro = self.read_off[lo:hi + 1].astype(np.int64)
        fo = self.ref_off[lo:hi + 1].astype(np.int64)
        return HostBatch(self.reads[ro[0]:ro[-1]].copy(), (ro - ro[0]).astype(np.uint32),
                         self.refs[fo[0]:fo[-1]].copy(), (fo - fo[0]).astype(np.uint32))

    def lengths(self) -> Tuple[np.ndarray, np.ndarray]:
        return np.diff(self.read_off.astype(np.int64)), np.diff(self.ref_off.astype(np.int64))

    @classmethod
    def read_seq_file(cls, path: str, max_pairs: Optional[int] = None) -> "HostBatch":
        """The harness's input format (benchmark_utils.h:325-352): line 2i = '>'+read, 2i+1 = '<'+ref; the
        first character of every line is skipped blindly."""
        pairs = []
        with open(path, "r") as fh:
            while max_pairs is None or len(pairs) < max_pairs:
                a = fh.readline()
                if not a:
                    break
                b = fh.readline()
                pairs.append((a.rstrip("\n")[1:], b.rstrip("\n")[1:]))
        return cls.from_strings(pairs)

    def write_seq_file(self, path: str) -> None:
        """benchmark_dataset.h:229,234 — '>%s\\n<%s\\n'."""
        with open(path, "w") as fh:
            for i in range(self.n):
                a, b = self.pair(i)
                fh.write(f">{a}\n<{b}\n")


def generate_pairs(cfg: GenConfig, first: int, n: int) -> HostBatch:
    """Host generator (asm_generate_pairs): pairs [first, first+n) of the seeded stream.  Needs no GPU."""
    lib = load_library()
    ro = np.zeros(n + 1, np.uint32)
    fo = np.zeros(n + 1, np.uint32)
    rc = lib.asm_generate_pairs(ctypes.byref(cfg), first, n, ro.ctypes.data, fo.ctypes.data, None, 0, None, 0)
    if rc:
        raise AsmError(rc, lib.asm_last_error(None).decode())
    reads = np.zeros(max(int(ro[-1]), 1), np.uint8)
    refs = np.zeros(max(int(fo[-1]), 1), np.uint8)
    rc = lib.asm_generate_pairs(ctypes.byref(cfg), first, n, ro.ctypes.data, fo.ctypes.data, reads.ctypes.data,
                                reads.size, refs.ctypes.data, refs.size)
    if rc:
        raise AsmError(rc, lib.asm_last_error(None).decode())
    return HostBatch(reads[:int(ro[-1])], ro, refs[:int(fo[-1])], fo)


class DeviceBatch:
    """asm_batch: a device-resident batch (ASCII + packed bit planes)."""

    def __init__(self, engine: "Engine", ptr: int):
        self.engine, self.ptr = engine, ptr
        self.n = int(engine.lib.asm_batch_size(ptr))
        self.max_length = int(engine.lib.asm_batch_max_length(ptr))
        self.ascii_bytes = int(engine.lib.asm_batch_text_bytes(ptr))

    def free(self) -> None:
        if self.ptr:
            self.engine.lib.asm_batch_free(self.engine.h, self.ptr)
            self.ptr = None

    def download(self) -> HostBatch:
        lib, h = self.engine.lib, self.engine.h
        ro = np.zeros(self.n + 1, np.uint32)
        fo = np.zeros(self.n + 1, np.uint32)
        self.engine._chk(lib.asm_batch_download(h, self.ptr, ro.ctypes.data, fo.ctypes.data, None, 0, None, 0))
        reads = np.zeros(max(int(ro[-1]), 1), np.uint8)
        refs = np.zeros(max(int(fo[-1]), 1), np.uint8)
        self.engine._chk(lib.asm_batch_download(h, self.ptr, None, None, reads.ctypes.data, reads.size,
                                                refs.ctypes.data, refs.size))
        return HostBatch(reads[:int(ro[-1])], ro, refs[:int(fo[-1])], fo)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Reference:
    """asm_reference: a reference text resident in HBM."""

    def __init__(self, engine: "Engine", ptr, length: int):
        self.engine, self.ptr, self.length = engine, ptr, length

    def free(self) -> None:
        if self.ptr:
            self.engine.lib.asm_reference_free(self.engine.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """asm_handle: one per GPU.  All device work of the hot path goes through here."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.asm_create(ctypes.byref(h), device)
        if rc:
            raise AsmError(rc, self.lib.asm_last_error(None).decode())
        self.h = h
        self.device = device

    def _chk(self, rc: int) -> None:
        if rc:
            raise AsmError(rc, self.lib.asm_last_error(self.h).decode())

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.asm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: Optional[int]) -> None:
        """Launch on a caller-owned hipStream_t.  0/None is HIP's legacy default stream (torch's `current_stream().cuda_stream`
        outside a stream context), not "the engine's own stream" — that is `reset_stream()`."""
        self._chk(self.lib.asm_set_stream(self.h, hip_stream or None))

    def reset_stream(self) -> None:
        self._chk(self.lib.asm_reset_stream(self.h))

    def synchronize(self) -> None:
        self._chk(self.lib.asm_synchronize(self.h))

    # ---- batches ----
    def upload(self, hb: HostBatch, greedy_mode: int = GREEDY_CLEAN) -> DeviceBatch:
        ptr = ctypes.c_void_p()
        reads = np.ascontiguousarray(hb.reads, np.uint8)
        refs = np.ascontiguousarray(hb.refs, np.uint8)
        ro = np.ascontiguousarray(hb.read_off, np.uint32)
        fo = np.ascontiguousarray(hb.ref_off, np.uint32)
        self._chk(self.lib.asm_batch_upload(self.h, hb.n, reads.ctypes.data, ro.ctypes.data, refs.ctypes.data,
                                            fo.ctypes.data, greedy_mode, ctypes.byref(ptr)))
        return DeviceBatch(self, ptr)

    def generate(self, cfg: GenConfig, first: int, n: int, greedy_mode: int = GREEDY_CLEAN) -> DeviceBatch:
        ptr = ctypes.c_void_p()
        self._chk(self.lib.asm_batch_generate(self.h, ctypes.byref(cfg), first, n, greedy_mode, ctypes.byref(ptr)))
        return DeviceBatch(self, ptr)

    def batch_from_text(self, text, greedy_mode: int = GREEDY_CLEAN) -> DeviceBatch:
        """asm_batch_from_text: a batch out of the harness's file format held in memory (bytes / uint8 array); the raw text
        goes to the GPU as it is and is parsed there."""
        buf = np.frombuffer(text.encode("ascii") if isinstance(text, str) else bytes(text), np.uint8) \
            if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
        ptr = ctypes.c_void_p()
        self._chk(self.lib.asm_batch_from_text(self.h, buf.ctypes.data if buf.size else None, buf.size, greedy_mode,
                                               ctypes.byref(ptr)))
        return DeviceBatch(self, ptr)

    def stream_seq_file(self, path: str, params: Params, greedy_mode: int = GREEDY_SEQUENTIAL,
                        aligners: Sequence[int] = (NW, LEAP, GREEDY), chunk_bytes: int = 0, max_pairs: int = 0,
                        capacity: Optional[int] = None, answers: Optional[np.ndarray] = None):
        """asm_stream_seq_file: `benchmark::read_string_file` + `run` (benchmark_utils.h:325-385) for a file of any size, streamed
        through pinned buffers with the parse, the pack and the aligners of one chunk overlapping the transfer of the next.
        -> (dict aligner -> int32[pairs], StreamStats).  capacity: entries of the result arrays (default: an upper bound from
        the file size; shorter reads need a larger bound, or max_pairs)."""
        if capacity is None:
            try:
                capacity = max_pairs if max_pairs > 0 else os.path.getsize(path) // 4 + 16
            except OSError:
                capacity = 16  # the library reports the unreadable file (benchmark_utils.h:350 only prints)
        mask = sum(1 << a for a in aligners)
        out = {a: np.zeros(capacity, np.int32) for a in aligners}
        st = StreamStats()
        ans = None if answers is None else np.ascontiguousarray(answers, np.int32)
        ptr = lambda a: out[a].ctypes.data if a in out else None  # noqa: E731
        self._chk(self.lib.asm_stream_seq_file(self.h, path.encode(), ctypes.byref(params), greedy_mode, mask, chunk_bytes, max_pairs,
                                               ptr(NW), ptr(LEAP), ptr(GREEDY), capacity, None if ans is None else ans.ctypes.data,
                                               0 if ans is None else ans.size, ctypes.byref(st)))
        return {a: v[:st.pairs] for a, v in out.items()}, st

    def upload_reference(self, text) -> "Reference":
        """Reference text (bytes / str / uint8 array) resident in HBM for seed-hit batches."""
        buf = np.frombuffer(text.encode("ascii") if isinstance(text, str) else bytes(text), np.uint8)
        ptr = ctypes.c_void_p()
        self._chk(self.lib.asm_reference_upload(self.h, buf.ctypes.data, buf.size, ctypes.byref(ptr)))
        return Reference(self, ptr, buf.size)

    def batch_from_hits(self, ref: "Reference", reads: np.ndarray, read_off: np.ndarray, hit_pos: np.ndarray,
                        greedy_mode: int = GREEDY_CLEAN) -> DeviceBatch:
        """The mapper's call shape (GASMA/mapper/main.cpp:77-86): pair i = (read i, reference window at hit_pos[i])."""
        reads = np.ascontiguousarray(reads, np.uint8)
        ro = np.ascontiguousarray(read_off, np.uint32)
        pos = np.ascontiguousarray(hit_pos, np.uint64)
        ptr = ctypes.c_void_p()
        self._chk(self.lib.asm_batch_from_hits(self.h, ref.ptr, ro.size - 1, reads.ctypes.data, ro.ctypes.data,
                                               pos.ctypes.data, greedy_mode, ctypes.byref(ptr)))
        return DeviceBatch(self, ptr)

    # ---- Greedy's sequential mode across batches (shards of one file / chunks of a stream) ----
    def tail_summary(self, batch: DeviceBatch) -> np.ndarray:
        """asm_batch_tail_summary: uint8[256], what this batch does to the reference's two persistent buffers."""
        out = np.zeros(256, np.uint8)
        self._chk(self.lib.asm_batch_tail_summary(self.h, batch.ptr, out.ctypes.data))
        return out

    def resolve_tails(self, batch: DeviceBatch, state: Optional[np.ndarray] = None) -> None:
        """asm_batch_resolve_tails: make `batch` a sequential-mode batch whose first pair sees `state` (uint8[256] of 2-bit
        codes; None = a file's start) in the buffers."""
        st = None if state is None else np.ascontiguousarray(state, np.uint8)
        if st is not None and st.size != 256:
            raise ValueError("state must have 256 entries")
        self._chk(self.lib.asm_batch_resolve_tails(self.h, batch.ptr, None if st is None else st.ctypes.data))

    def pack_async(self, batch: DeviceBatch) -> None:
        self._chk(self.lib.asm_batch_pack_async(self.h, batch.ptr))

    # ---- device memory ----
    def malloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._chk(self.lib.asm_device_malloc(self.h, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, ptr: int) -> None:
        self._chk(self.lib.asm_device_free(self.h, ptr))

    def to_host(self, ptr: int, count: int, dtype=np.int32) -> np.ndarray:
        out = np.zeros(count, dtype)
        self._chk(self.lib.asm_memcpy_d2h(self.h, out.ctypes.data, ptr, out.nbytes))
        return out

    def memset_async(self, ptr: int, value: int, nbytes: int) -> None:
        self._chk(self.lib.asm_memset_async(self.h, ptr, value, nbytes))

    # ---- the hot path ----
    def align_async(self, batch: DeviceBatch, aligner: int, params: Params, d_out: int) -> None:
        """One aligner over a resident batch into a device int32[n] buffer (enqueue only)."""
        self._chk(self.lib.asm_align_batch_async(self.h, batch.ptr, aligner, ctypes.byref(params), d_out))

    def align_hinted_async(self, batch: DeviceBatch, aligner: int, params: Params, d_hint: Optional[int], d_out: int) -> None:
        """align_async with a per-pair work estimate (device int32[n]) that only steers scheduling."""
        self._chk(self.lib.asm_align_batch_hinted_async(self.h, batch.ptr, aligner, ctypes.byref(params), d_hint, d_out))

    def align(self, batch: DeviceBatch, aligner: int, params: Params) -> np.ndarray:
        d_out = self.malloc(4 * max(batch.n, 1))
        try:
            self.align_async(batch, aligner, params, d_out)
            return self.to_host(d_out, batch.n)
        finally:
            self.free(d_out)

    # ---- filtering stage: bit-parallel LEAP (SIMD_ED) and SHD (LEAP_SIMD/main.cpp:95-101,186-195) ----
    def simd_ed_async(self, batch: DeviceBatch, ed_threshold: int, d_ed: int, shd: bool = True, mode: int = FILTER_CLEAN,
                      state: Optional[Sequence[int]] = None, ed_mode: int = 0) -> Optional[Tuple[int, ...]]:
        """SIMD_ED::init_levenshtein(ed_threshold, ED_GLOBAL, shd) + load_reads/calculate_masks/reset/run per pair:
        d_ed[i] = get_ED() when check_pass() else -1.  state = (final_ED, lane distance, converge_ED) carried into the
        first pair in FILTER_SEQUENTIAL mode; returns the state after the last pair (for the next chunk of the same file).
        ed_mode: init_levenshtein's ED_modes (LEAP_GLOBAL, the filter driver's; LEAP_LOCAL / LEAP_SEMI_FREE_BEGIN / LEAP_SEMI_FREE_END)."""
        st = None
        if state is not None:
            st = (ctypes.c_int32 * 3)(*[int(v) for v in state])
        self._chk(self.lib.asm_simd_ed_mode_batch_async(self.h, batch.ptr, int(ed_threshold), 1 if shd else 0, int(mode), int(ed_mode),
                                                        st, d_ed))
        return tuple(st) if st is not None else None

    def simd_ed(self, batch: DeviceBatch, ed_threshold: int, shd: bool = True, mode: int = FILTER_CLEAN,
                state: Optional[Sequence[int]] = None, ed_mode: int = 0) -> np.ndarray:
        d_out = self.malloc(4 * max(batch.n, 1))
        try:
            self.simd_ed_async(batch, ed_threshold, d_out, shd, mode, state, ed_mode)
            return self.to_host(d_out, batch.n)
        finally:
            self.free(d_out)

    def simd_ed_affine_async(self, batch: DeviceBatch, gap_threshold: int, af_threshold: int, x: int, o: int, e: int, d_ed: int,
                             shd_threshold: Optional[int] = None, mode: int = 0) -> None:
        """SIMD_ED::init_affine(gap_threshold, af_threshold, ED_GLOBAL, x, o, e[, true, shd_threshold]) + load_reads/calculate_masks/
        reset/run per pair, every pair from clean tables: d_ed[i] = get_ED() when check_pass() (1000000 for a pair exact at e = 0)
        else -1.  shd_threshold: init_affine's SHD_enable = true with that SHD_threshold (None: off, the reference's default)."""
        if mode != 0:  # init_affine's ED_modes (LEAP_LOCAL / LEAP_SEMI_FREE_BEGIN / LEAP_SEMI_FREE_END)
            self._chk(self.lib.asm_simd_ed_affine_mode_batch_async(self.h, batch.ptr, int(gap_threshold), int(af_threshold), int(x),
                                                                   int(o), int(e), -1 if shd_threshold is None else int(shd_threshold),
                                                                   int(mode), d_ed))
        elif shd_threshold is None:
            self._chk(self.lib.asm_simd_ed_affine_batch_async(self.h, batch.ptr, int(gap_threshold), int(af_threshold), int(x), int(o),
                                                              int(e), d_ed))
        else:
            self._chk(self.lib.asm_simd_ed_affine_shd_batch_async(self.h, batch.ptr, int(gap_threshold), int(af_threshold), int(x),
                                                                  int(o), int(e), int(shd_threshold), d_ed))

    def simd_ed_affine(self, batch: DeviceBatch, gap_threshold: int, af_threshold: int, x: int, o: int, e: int,
                       shd_threshold: Optional[int] = None, mode: int = 0) -> np.ndarray:
        d_out = self.malloc(4 * max(batch.n, 1))
        try:
            self.simd_ed_affine_async(batch, gap_threshold, af_threshold, x, o, e, d_out, shd_threshold, mode)
            return self.to_host(d_out, batch.n)
        finally:
            self.free(d_out)

    def shd_filter_async(self, batch: DeviceBatch, max_error: int, d_pass: int) -> None:
        """bit_vec_filter_avx(read planes, ref planes, length, max_error): d_pass[i] in {0, 1}."""
        self._chk(self.lib.asm_shd_filter_batch_async(self.h, batch.ptr, int(max_error), d_pass))

    def shd_filter(self, batch: DeviceBatch, max_error: int) -> np.ndarray:
        d_out = self.malloc(4 * max(batch.n, 1))
        try:
            self.shd_filter_async(batch, max_error, d_out)
            return self.to_host(d_out, batch.n)
        finally:
            self.free(d_out)

    def align_host(self, hb: HostBatch, aligner: int, params: Params, greedy_mode: int = GREEDY_CLEAN) -> np.ndarray:
        """asm_align_batch: host in, host out — `align(read, ref, k)` for every pair of the batch."""
        out = np.zeros(hb.n, np.int32)
        reads = np.ascontiguousarray(hb.reads, np.uint8)
        refs = np.ascontiguousarray(hb.refs, np.uint8)
        ro = np.ascontiguousarray(hb.read_off, np.uint32)
        fo = np.ascontiguousarray(hb.ref_off, np.uint32)
        self._chk(self.lib.asm_align_batch(self.h, aligner, hb.n, reads.ctypes.data, ro.ctypes.data, refs.ctypes.data,
                                           fo.ctypes.data, ctypes.byref(params), greedy_mode, out.ctypes.data))
        return out

    def greedy_with_cigar(self, batch: DeviceBatch, params: Params, cap: int = 48):
        """-> (costs int32[n], CIGAR strings) — hurdle_matrix::get_cost / get_CIGAR for every pair of the batch."""
        n = batch.n
        d_pen, d_ops, d_nops = self.malloc(4 * max(n, 1)), self.malloc(2 * cap * max(n, 1)), self.malloc(max(n, 1))
        try:
            self._chk(self.lib.asm_greedy_cigar_batch_async(self.h, batch.ptr, ctypes.byref(params), d_pen, d_ops, cap,
                                                            d_nops))
            costs = self.to_host(d_pen, n)
            ops = self.to_host(d_ops, n * cap, np.uint16).reshape(n, cap)
            nops = self.to_host(d_nops, n, np.uint8)
        finally:
            for p in (d_pen, d_ops, d_nops):
                self.free(p)
        return costs, decode_cigars(ops, nops, cap), nops

    def coverage(self, batch: DeviceBatch, params: Params, window: int = 64, cap: int = 64, want_nw_cigars: bool = False):
        """The harness's coverage metric for every pair: -> dict(cover uint8[n] (1/0/2), covered, undetermined,
        greedy_cost, greedy_cigars[, nw_cigars])."""
        n = batch.n
        d_pen, d_ops, d_nops = self.malloc(4 * max(n, 1)), self.malloc(2 * cap * max(n, 1)), self.malloc(max(n, 1))
        d_cov, d_cnt = self.malloc(max(n, 1)), self.malloc(16)
        d_nwo = self.malloc(2 * cap * max(n, 1)) if want_nw_cigars else None
        d_nwn = self.malloc(max(n, 1)) if want_nw_cigars else None
        try:
            self._chk(self.lib.asm_greedy_cigar_batch_async(self.h, batch.ptr, ctypes.byref(params), d_pen, d_ops, cap,
                                                            d_nops))
            self.memset_async(d_cnt, 0, 16)
            self._chk(self.lib.asm_coverage(self.h, batch.ptr, ctypes.byref(params), d_ops, cap, d_nops, window, d_cov,
                                            d_nwo, cap, d_nwn, d_cnt))
            cnt = self.to_host(d_cnt, 2, np.uint64)
            out = {"cover": self.to_host(d_cov, n, np.uint8), "covered": int(cnt[0]), "undetermined": int(cnt[1]),
                   "greedy_cost": self.to_host(d_pen, n)}
            gops = self.to_host(d_ops, n * cap, np.uint16).reshape(n, cap)
            out["greedy_cigars"] = decode_cigars(gops, self.to_host(d_nops, n, np.uint8), cap)
            if want_nw_cigars:
                nops = self.to_host(d_nwn, n, np.uint8)
                out["nw_cigars"] = decode_cigars(self.to_host(d_nwo, n * cap, np.uint16).reshape(n, cap), nops, cap,
                                                 reverse=True)
            return out
        finally:
            for p in (d_pen, d_ops, d_nops, d_cov, d_cnt, d_nwo, d_nwn):
                if p:
                    self.free(p)

    def count_equal_async(self, d_a: int, d_b: int, n: int, d_count: int) -> None:
        self._chk(self.lib.asm_count_equal_async(self.h, d_a, d_b, n, d_count))

    def accuracy_async(self, d_nw: int, d_leap: Optional[int], d_greedy: Optional[int], n: int, d_counters: int,
                       d_answers: Optional[int] = None) -> None:
        """counters (uint64[4]) += {total, nw_ok, leap_ok, greedy_ok} — benchmark_utils.h:249-255."""
        self._chk(self.lib.asm_accuracy_async(self.h, d_nw, d_leap, d_greedy, d_answers, n, d_counters))

    def run_benchmark_async(self, batch: DeviceBatch, params: Params, d_nw: Optional[int], d_leap: Optional[int],
                            d_greedy: Optional[int], d_counters: Optional[int], repack: bool = True,
                            d_answers: Optional[int] = None) -> None:
        """`_run_benchmark` over the whole resident batch (pack, NW, LEAP, Greedy, counters) in one call.  repack: False / True
        (pack in stream order) / 2 (pipelined: this call's pack overlaps the previous call's aligners) / 3 (overlapped calls:
        alternate two sets of output arrays from call to call and end with pipeline_join_async)."""
        self._chk(self.lib.asm_run_benchmark_async(self.h, batch.ptr, ctypes.byref(params), int(repack), d_nw,
                                                   d_leap, d_greedy, d_answers, d_counters))

    def pipeline_join_async(self) -> None:
        """Orders everything the overlapped calls (repack=3) enqueued before what comes next on the handle's stream."""
        self._chk(self.lib.asm_pipeline_join_async(self.h))

    # ---- timing ----
    def profile_enable(self, max_calls: int, kernel_mask: int = 0xF) -> None:
        """The next max_calls run_benchmark_async calls time the selected kernels (bit 0 pack, 1 NW, 2 LEAP, 3 Greedy) with
        events on the launching streams."""
        self._chk(self.lib.asm_profile_enable(self.h, int(max_calls), int(kernel_mask)))

    def profile_read(self, cap_calls: int) -> np.ndarray:
        """-> float32[calls][4] = ms of (pack, nw, leap, greedy) per recorded call, -1 where not launched; synchronises."""
        ms = np.full((max(cap_calls, 1), 4), -1.0, np.float32)
        n = ctypes.c_int(0)
        self._chk(self.lib.asm_profile_read(self.h, ms.ctypes.data, int(cap_calls), ctypes.byref(n)))
        return ms[:min(n.value, cap_calls)]

    def timer(self) -> "Timer":
        return Timer(self)


def decode_cigars(ops: np.ndarray, nops: np.ndarray, cap: int, reverse: bool = False):
    """Encoded rows (count << 3 | op; op 0 'M', 1 'I', 2 'D', 3 '=', 4 'X') -> CIGAR strings."""
    letters = "MID=X???"
    out = []
    for i in range(ops.shape[0]):
        row = ops[i, :min(int(nops[i]), cap)]
        if reverse:
            row = row[::-1]
        out.append("".join(f"{int(v) >> 3}{letters[int(v) & 7]}" for v in row))
    return out


class Timer:
    """HIP events on the engine's stream."""

    def __init__(self, engine: Engine):
        self.e = engine
        self.t = ctypes.c_void_p()
        engine._chk(engine.lib.asm_timer_create(engine.h, ctypes.byref(self.t)))

    def start(self) -> None:
        self.e._chk(self.e.lib.asm_timer_start(self.e.h, self.t))

    def stop(self) -> None:
        self.e._chk(self.e.lib.asm_timer_stop(self.e.h, self.t))

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        self.e._chk(self.e.lib.asm_timer_elapsed_ms(self.e.h, self.t, ctypes.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            self.e.lib.asm_timer_destroy(self.e.h, self.t)
        except Exception:
            pass


def declared_symbols() -> Sequence[str]:
    """Function names declared in include/asm_mi355x.h (parsed from the header text)."""
    import re

    with open(HEADER_PATH) as fh:
        text = fh.read()
    return sorted(set(re.findall(r"\b(asm_[a-z0-9_]+)\s*\(", text)) - {"asm_handle", "asm_batch"})


# ---- multi-GPU: read pairs are independent, so the batch shards with no data-path collective ----
def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Strong-scaling split of `total` pairs into contiguous blocks of ceil(total/world): -> [lo, hi)."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def weak_shard_first(rank: int, pairs_per_rank: int) -> int:
    """Weak scaling: rank r owns pairs [r*n, (r+1)*n) of the seeded stream."""
    return rank * pairs_per_rank


def tail_state_advance(state: np.ndarray, summary: np.ndarray, n_pairs: int) -> np.ndarray:
    """asm_tail_state_advance (host only, needs no GPU): buffer state after a batch of n_pairs with the given summary."""
    lib = load_library()
    st = np.ascontiguousarray(state, np.uint8).copy()
    sm = np.ascontiguousarray(summary, np.uint8)
    rc = lib.asm_tail_state_advance(st.ctypes.data, sm.ctypes.data, int(n_pairs))
    if rc:
        raise AsmError(rc, lib.asm_last_error(None).decode())
    return st


def chain_tail_state(summary: np.ndarray, n_pairs: int, dist=None, device=None) -> np.ndarray:
    """The buffer state before THIS rank's shard of a file whose shards are laid out in rank order: one all-gather of every
    shard's 256-byte summary and size (the only exchange sequential mode needs), then a local fold over the shards before
    ours.  Rank 0, or no process group: zeros — a file's start (hurdle_matrix.h:136-137)."""
    state = np.zeros(256, np.uint8)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return state
    import torch

    mine = torch.zeros(264, dtype=torch.uint8, device=device)
    mine[:256] = torch.from_numpy(np.ascontiguousarray(summary, np.uint8)).to(mine.device)
    mine[256:] = torch.from_numpy(np.frombuffer(np.int64(n_pairs).tobytes(), np.uint8).copy()).to(mine.device)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    for r in range(dist.get_rank()):
        row = parts[r].cpu().numpy()
        state = tail_state_advance(state, row[:256], int(np.frombuffer(row[256:].tobytes(), np.int64)[0]))
    return state


def allreduce_counters(counters, dist=None):
    """The one collective of the path: sum the {total, nw_ok, leap_ok, greedy_ok} int64 counters over ranks
    (RCCL over xGMI on GPUs — backend "nccl"; gloo in the CPU tests).  `counters` is a torch tensor."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counters)
    return counters


def gather_penalties(local, dist=None, dst: int = 0):
    """Optional: collect the per-shard penalty tensors on rank `dst` (each peer sends over its own direct link; shards
    may differ in length by one pair).  Returns the list of shards on `dst`, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    import torch

    world = dist.get_world_size()
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.numel()], dtype=torch.int64, device=local.device))
    longest = max(int(s.item()) for s in sizes)
    padded = torch.zeros(longest, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    bufs = [torch.empty(longest, dtype=local.dtype, device=local.device) for _ in range(world)] \
        if dist.get_rank() == dst else None
    dist.gather(padded, bufs, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [b[:int(s.item())] for b, s in zip(bufs, sizes)]
