"""Host-side plumbing over the C ABI: NHWC activation views, filter packing
(BatchNorm folding, OIHW -> OHWI, padding) and thin op wrappers.

PyTorch is used only as the device allocator / stream provider; every compute
call goes through ``libfcp_hip.so``.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as N
from . import torch_ops as T

BN_EPS = 1e-5
COUT_PAD = 128  # filters are zero-padded to a multiple of the largest N tile


class Act:
    """A channel-slice view of an NHWC fp32 device buffer.

    ``buf`` has shape (n, h, w, ld); the view covers channels [c0, c0+c).  This
    is how ``torch.cat`` along channels is realised without a copy: producers
    write their slice, consumers read a wider slice.
    """
    __slots__ = ("buf", "c0", "c", "fmt")

    def __init__(self, buf: torch.Tensor, c0: int = 0, c: int | None = None, fmt: int = 0):
        """``fmt``: 0 = fp32 NHWC; 1 = "split32" (same bytes: per 32 channels, 32 binary16 hi parts then
        32 binary16 lo parts; value = hi + lo) — the operand image of the fp16x3 conv kernel."""
        assert buf.dim() == 4 and buf.dtype == torch.float32 and buf.is_contiguous()
        self.buf, self.c0, self.fmt = buf, c0, fmt
        self.c = buf.shape[3] - c0 if c is None else c
        assert 0 <= c0 and c0 + self.c <= buf.shape[3]
        if fmt == 1:
            assert c0 % 32 == 0 and self.c % 32 == 0 and buf.shape[3] % 32 == 0, "split32 views are 32-channel aligned"

    @staticmethod
    def empty(n, h, w, c, device, fmt: int = 0):
        return Act(torch.empty((n, h, w, c), dtype=torch.float32, device=device), fmt=fmt)

    @property
    def n(self): return self.buf.shape[0]
    @property
    def h(self): return self.buf.shape[1]
    @property
    def w(self): return self.buf.shape[2]
    @property
    def ld(self): return self.buf.shape[3]

    def slice(self, c0, c):
        return Act(self.buf, self.c0 + c0, c, self.fmt)

    def ptr(self):
        return N.ptr(self.buf, 4 * self.c0)

    def nchw(self) -> torch.Tensor:
        """Copy out as an fp32 NCHW torch tensor (tests only)."""
        v = self.buf[..., self.c0:self.c0 + self.c].contiguous()
        if self.fmt == 1:
            v = split32_to_f32(Act(v, fmt=1)).buf
        return v.permute(0, 3, 1, 2).contiguous()


@dataclass
class PackedConv:
    w: torch.Tensor          # packed filter on device
    bias: torch.Tensor | None
    cin: int
    cout: int
    kh: int
    kw: int
    stride: int
    pad: int
    cin4: bool
    flops_per_pixel: int     # 2*cout*cin*kh*kw (algorithmic, unpadded)
    precision: int = 0       # 0 fp32 exact, 1 fp16x3 split
    wscale: torch.Tensor | None = None


def fold_bn(weight: np.ndarray, bn: dict | None, bias: np.ndarray | None):
    """conv -> BN(eval) folded into (weight, bias), fp32 like ATen's inference BN:
    alpha = gamma / sqrt(var + eps); y = conv*alpha + (beta - mean*alpha)."""
    w = np.asarray(weight, dtype=np.float32)
    if bn is None:
        return w, (None if bias is None else np.asarray(bias, np.float32))
    f = np.float32
    alpha = (bn["weight"].astype(f) / np.sqrt(bn["running_var"].astype(f) + f(BN_EPS))).astype(f)
    beta = (bn["bias"].astype(f) - bn["running_mean"].astype(f) * alpha).astype(f)
    if bias is not None:
        beta = (beta + np.asarray(bias, f) * alpha).astype(f)
    return (w * alpha[:, None, None, None]).astype(f), beta


PRECISIONS = {"f32": 0, "fp32": 0, "exact": 0, 0: 0, "f16x3": 1, "fp16x3": 1, 1: 1}
# module-wide default for pack_conv(precision=None): env FCP_PRECISION = f32 | f16x3
DEFAULT_PRECISION = PRECISIONS[os.environ.get("FCP_PRECISION", "f16x3")]


def resolve_precision(p):
    if p is None:
        return DEFAULT_PRECISION
    if p not in PRECISIONS:
        raise ValueError(f"unknown precision {p!r}: use 'f32' (exact fp32 MFMA) or 'f16x3' (split fp16 MFMA)")
    return PRECISIONS[p]


@contextlib.contextmanager
def default_precision(p):
    """Temporarily set the precision ``pack_conv`` uses when none is given."""
    global DEFAULT_PRECISION
    prev, DEFAULT_PRECISION = DEFAULT_PRECISION, resolve_precision(p)
    try:
        yield
    finally:
        DEFAULT_PRECISION = prev



def split_f16x3(packed: np.ndarray):
    """fp32 packed filter [cout_pad][K] -> (uint16 image with, per 32 K values, 32 hi + 32 lo
    binary16 numbers, wscale).  Rows are pre-scaled by 2^-floor(log2(max|w|)) so the lo parts stay
    normal in binary16; ``wscale`` (= 2^e per row) undoes the scaling exactly in the epilogue."""
    rows = packed.reshape(packed.shape[0], -1).astype(np.float32)
    amax = np.abs(rows).max(1)
    e = np.where(amax > 0, np.floor(np.log2(np.maximum(amax, 1e-38))), 0.0)
    scale = np.exp2(e).astype(np.float32)
    ws = rows / scale[:, None]
    hi = ws.astype(np.float16)
    lo = (ws - hi.astype(np.float32)).astype(np.float16)
    k = rows.shape[1]
    assert k % 32 == 0
    img = np.empty((rows.shape[0], k // 32, 2, 32), np.float16)
    img[:, :, 0] = hi.reshape(rows.shape[0], k // 32, 32)
    img[:, :, 1] = lo.reshape(rows.shape[0], k // 32, 32)
    return img.reshape(rows.shape[0], k * 2).view(np.uint16), scale


def pack_conv(weight, bias=None, bn=None, stride=1, pad=0, device="cuda", cin_perm=None,
              precision: int | None = None) -> PackedConv:
    """weight: (cout,cin,kh,kw) numpy/torch; bn: dict with weight/bias/running_mean/running_var.
    ``precision``: 0 = exact fp32 MFMA, 1 = fp16x3 split MFMA (None: ``DEFAULT_PRECISION``)."""
    precision = resolve_precision(precision)
    if isinstance(weight, torch.Tensor):
        weight = weight.detach().cpu().numpy()
    if isinstance(bias, torch.Tensor):
        bias = bias.detach().cpu().numpy()
    if bn is not None:
        bn = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in bn.items()}
    w, b = fold_bn(weight, bn, bias)
    if cin_perm is not None:
        w = w[:, cin_perm]
    cout, cin, kh, kw = w.shape
    cout_pad = -(-cout // COUT_PAD) * COUT_PAD
    cin4 = cin <= 4
    if cin4:
        assert kw <= 8
        packed = np.zeros((cout_pad, kh, 8, 4), np.float32)
        packed[:cout, :, :kw, :cin] = w.transpose(0, 2, 3, 1)
    else:
        assert cin % 32 == 0, f"cin={cin} must be a multiple of 32"
        # K order (cin/32, kh, kw, 32): the taps of one 32-channel slice are consecutive
        packed = np.zeros((cout_pad, cin // 32, kh, kw, 32), np.float32)
        packed[:cout] = w.transpose(0, 2, 3, 1).reshape(cout, kh, kw, cin // 32, 32).transpose(0, 3, 1, 2, 4)
    wsc = None
    if precision == 1:
        img, scale = split_f16x3(packed)
        wd = torch.from_numpy(np.ascontiguousarray(img.view(np.int16))).to(device)
        wsc = torch.from_numpy(scale).to(device)
    else:
        wd = torch.from_numpy(np.ascontiguousarray(packed)).to(device)
    bd = None if b is None else torch.from_numpy(np.ascontiguousarray(b)).to(device)
    return PackedConv(wd, bd, 4 if cin4 else cin, cout, kh, kw, stride, pad, cin4,
                      2 * cout * cin * kh * kw, precision, wsc)


def bn_of(sd, prefix):
    return {k: sd[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}


def _pick_tile_n(cout: int, m: int) -> int:
    """Largest N tile that wastes no matrix work and still fills 256 CUs."""
    if cout <= 32:
        return 32
    gm = -(-m // 128)
    if cout % 128 == 0 and gm * (cout // 128) >= 512:
        return 128
    return 64


_props_lock = threading.Lock()
_props: dict = {}


def device_props(dev=None):
    """``torch.cuda.get_device_properties`` behind a lock and a cache.  process_dir's GPU worker threads reach their first
    launch at the same time, and concurrent first calls into torch's lazy device bookkeeping raced on the MI355X boxes
    ("AssertionError: Invalid device id" from a worker thread, then an abort at teardown)."""
    idx = torch.cuda.current_device() if dev is None else (torch.device(dev).index if torch.device(dev).index is not None
                                                           else torch.cuda.current_device())
    got = _props.get(idx)
    if got is None:
        with _props_lock:
            got = _props.get(idx)
            if got is None:
                got = _props[idx] = torch.cuda.get_device_properties(idx)
    return got


class _StreamClaim:
    """Held in a host thread's thread-local storage: gives the thread's slot back when the thread ends."""

    def __init__(self, dev_index, slot):
        self.dev_index, self.slot = dev_index, slot

    def __del__(self):
        try:
            with _stream_lock:
                _stream_free.setdefault(self.dev_index, set()).add(self.slot)
        except Exception:                                    # interpreter shutdown
            pass


_stream_lock = threading.RLock()    # re-entrant: a _StreamClaim may be finalised while its thread holds the lock
_stream_sets: dict = {}          # device index -> [slot -> {"main": Stream | None, "side": {k: [k Streams]}}]
_stream_free: dict = {}          # device index -> slots no living thread holds
_stream_tls = threading.local()


def thread_streams(dev) -> dict:
    """The calling host thread's persistent set of HIP streams on ``dev``: ``{"main": ..., "side": {k: [...]}}`` (both filled
    lazily by ``thread_main_stream`` / ``thread_side_streams``).  Sets live in process-wide SLOTS: a thread claims the lowest
    free slot at its first call and gives it back when it ends, so the GPU worker threads of successive ``process_dir``
    runs — and every detector object a thread runs — get the SAME streams again instead of fresh ones.  Why that matters
    (profiles/r06_probes.md section 2): HIP multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues, and the side streams of a
    second, freshly created pair were measured sharing one queue: the detector's two half-batches then run one after the other
    (17.9 -> 20.3 ms per batch-64 step) — every `extra` record of bench.py paid that until round 6."""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    claims = _stream_tls.__dict__.setdefault("claims", {})
    claim = claims.get(idx)
    with _stream_lock:
        sets = _stream_sets.setdefault(idx, [])
        if claim is None:
            free = _stream_free.setdefault(idx, set())
            slot = min(free) if free else len(sets)
            free.discard(slot)
            if slot == len(sets):
                sets.append({"main": None, "side": {}})
            claim = claims[idx] = _StreamClaim(idx, slot)
        return sets[claim.slot]


STREAM_PROBE = os.environ.get("FCP_STREAM_PROBE", "1") != "0"
_stream_rejects: list = []       # streams that shared a queue with a chosen one: kept alive (their queue assignment stands)


def _streams_overlap(a, b, cycles: int = 1_500_000) -> bool:
    """Whether kernels on streams ``a`` and ``b`` run concurrently: a one-workgroup spin kernel (``torch.cuda._sleep``) on each;
    two that share a hardware queue take twice as long as one (tools/probe_hw_queues.py: 1.0 x against 2.0 x, nothing between)."""
    import time

    def timed(streams):
        for s_ in streams:
            s_.synchronize()
        t = time.perf_counter()
        for s_ in streams:
            with torch.cuda.stream(s_):
                torch.cuda._sleep(cycles)
        for s_ in streams:
            s_.synchronize()
        return time.perf_counter() - t
    timed([a])                                               # first use: the runtime assigns the stream's hardware queue here
    timed([b])
    one = min(timed([a]), timed([b]))
    return min(timed([a, b]), timed([a, b])) < 1.5 * one


def _new_stream_beside(dev, others, tries: int = 6):
    """A new stream whose kernels overlap with those of every stream in ``others``.  HIP assigns a stream to one of
    GPU_MAX_HW_QUEUES (4) hardware queues at its first use — the least loaded one, so the 3rd and 4th streams of a process were
    measured on ONE queue while the 1st / 2nd and 5th / 6th were on two (tools/probe_hw_queues.py, profiles/r06_probes.md section
    2) — and two streams on one queue run their kernels one after the other.  Candidates that alias a chosen stream are set
    aside (alive, so the next candidate lands elsewhere); after ``tries`` candidates the last one is taken as it is (more
    streams than queues: some sharing is unavoidable, and only costs overlap)."""
    with torch.cuda.device(dev):
        cand = torch.cuda.Stream(device=dev)
        if not STREAM_PROBE or not others or not hasattr(torch.cuda, "_sleep") or torch.cuda.is_current_stream_capturing():
            return cand
        for _ in range(tries):
            try:
                if all(_streams_overlap(o, cand) for o in others):
                    return cand
            except Exception:                                # a probe must never break the data path
                return cand
            _stream_rejects.append(cand)
            cand = torch.cuda.Stream(device=dev)
        return cand


def thread_side_streams(dev, k: int, with_main: bool = False):
    """k streams for the detector's k sub-batches (the heavy streams) of the calling thread, on hardware queues of their own.
    ``with_main``: the thread's main stream is the first of the k (a GPU worker thread of ``process_dir`` that already runs on
    its main stream: one sub-batch stays there, k - 1 side streams are added).  Measured with two GPU workers
    (tools/bench_process_dir.py with FCP_STREAM_MATRIX=1, profiles/r06_probes.md section 2): when a worker's main stream
    shared a queue with the OTHER worker's heavy stream — its decode / NMS / warp launches then queue behind the other
    worker's convolutions — process_dir ran at 2.3-2.5 k images/s; with every main stream on the queue of its own sub-batch
    3.0-3.3 k.  Only the thread's own streams are kept apart from each other; other threads' streams may share their queues."""
    st = thread_streams(dev)
    key = (k, bool(with_main))
    if key not in st["side"]:
        base = [thread_main_stream(dev)] if with_main else []
        with _stream_lock:                                   # one prober at a time
            side = list(base)
            while len(side) < k:
                side.append(_new_stream_beside(dev, side))
        st["side"][key] = side
    return st["side"][key]


def thread_main_stream(dev):
    """The stream a GPU worker thread of ``process_dir`` runs its batches on (light work — fork / join of the side streams,
    post-processing, align: it may share a queue with anything)."""
    st = thread_streams(dev)
    if st["main"] is None:
        with torch.cuda.device(dev):
            st["main"] = torch.cuda.Stream(device=dev)
    return st["main"]


BIG_TILES = os.environ.get("FCP_BIG_TILES", "1") != "0"   # offer the 256-row kernel to the autotuner
BALANCE_TAIL = os.environ.get("FCP_BALANCE_TAIL", "1") != "0"   # and its balanced M-tile schedule (FCP_CONV_BALANCE_TAIL)

_budget = threading.local()


@contextlib.contextmanager
def cu_budget(cus: int):
    """CUs the 256-row conv launches of this thread may count on when they lay out their last dispatch round (0 = the
    whole device).  A caller that runs k sub-batches concurrently on k streams passes 1/k of the device."""
    prev = getattr(_budget, "cus", 0)
    _budget.cus = int(cus)
    try:
        yield
    finally:
        _budget.cus = prev


class Autotune:
    """Optional per-shape choice of the N tile: the first launch of an unseen conv shape times the
    candidate tiles with HIP events (costs a sync and a handful of extra launches per new shape) and caches the
    winner; every candidate of the fp16x3 path returns bit-identical tensors, so the choice never changes a
    result.  On by default (env FCP_AUTOTUNE=0 turns it off); a shape tuned once keeps its tile afterwards."""
    enabled = os.environ.get("FCP_AUTOTUNE", "1") != "0"
    # FCP_AUTOTUNE=0 suppresses the tuning LAUNCHES only (determinism of the launch sequence, start-up latency): shapes the
    # shipped / user tables know keep their tuned tile.  Ignoring the tables as well — the heuristic tiles, what
    # ``bench.py --no-autotune`` measures — is its own switch: FCP_TUNE_TABLES=0.
    use_tables = os.environ.get("FCP_TUNE_TABLES", "1") != "0"
    cache: dict = {}
    _lock = threading.Lock()     # process_dir's GPU workers share the cache: one tuner at a time
    # Picks persist on disk, keyed by (ISA name, CU count, ABI version) + shape: a second start skips the timing
    # launches.  Read order: the table shipped in face-crop-plus_amd/tuned/ (picks measured on an MI355X by
    # tools/dump_autotune.py), then the user's file ($FCP_TUNE_CACHE, default ~/.cache/face_crop_plus_amd/autotune.json;
    # "0" = no disk cache at all), which also receives every new pick.  Every candidate returns the same bits, so a
    # stale pick can only cost speed; one that is no longer among a shape's candidates is ignored.
    # Version of the tile vocabulary the tables are keyed by (the ABI version at which the candidate set last changed its
    # meaning; an ABI bump that only adds descriptor fields keeps the tables valid)
    TABLE_VERSION = 13
    _disk_loaded = False
    _disk_section = None         # "<device name>|<CUs>|abi<N>"
    _disk_dirty = False

    @classmethod
    def _user_path(cls):
        p = os.environ.get("FCP_TUNE_CACHE")
        if p == "0":
            return None
        return p or os.path.join(os.path.expanduser("~"), ".cache", "face_crop_plus_amd", "autotune.json")

    @classmethod
    def section(cls):
        if cls._disk_section is None:
            prop = device_props()
            # the marketing name differs between boxes of one pool ("AMD Radeon Graphics" / "AMD Instinct MI355X"):
            # the ISA name + CU count identify the part
            arch = getattr(prop, "gcnArchName", prop.name).split(":")[0]
            cls._disk_section = f"{arch}|{prop.multi_processor_count}|abi{cls.TABLE_VERSION}"
        return cls._disk_section

    @classmethod
    def ensure_loaded(cls):
        """Merge the shipped and the user's tables into ``cache`` (once per process; in-process picks win)."""
        if cls._disk_loaded or not cls.use_tables:
            return
        with cls._lock:
            if cls._disk_loaded:
                return
            cls._disk_loaded = True
            if os.environ.get("FCP_TUNE_CACHE") == "0":
                return
            import ast
            import json
            shipped = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned")
            files = sorted(os.path.join(shipped, f) for f in os.listdir(shipped) if f.endswith(".json")) if os.path.isdir(shipped) else []
            if cls._user_path() and os.path.isfile(cls._user_path()):
                files.append(cls._user_path())
            try:
                sec = cls.section()
            except Exception:                                # no GPU: nothing to key on
                return
            for path in files:
                try:
                    with open(path) as f:
                        table = json.load(f).get(sec, {})
                    for k, v in table.items():
                        cls.cache.setdefault(ast.literal_eval(k), tuple(v))
                except Exception as e:                       # a damaged table costs a re-tune, never a failure
                    import warnings
                    warnings.warn(f"autotune table {path} ignored: {type(e).__name__}: {e}")

    @classmethod
    def save(cls, path: str | None = None):
        """Write this process's picks into the user's table (merged with what the file holds for other devices / shapes)."""
        path = path or cls._user_path()
        if path is None:
            return None
        import json
        with cls._lock:
            table = {}
            try:
                with open(path) as f:
                    table = json.load(f)
            except Exception:
                pass
            sec = table.setdefault(cls.section(), {})
            for k, v in list(cls.cache.items()):
                sec[repr(k)] = list(v)
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump(table, f, indent=0, sort_keys=True)
            os.replace(tmp, path)
            cls._disk_dirty = False
        return path

    @classmethod
    def pick(cls, key, candidates, launch):
        best = cls.cache.get(key)
        if best is not None:
            return best
        with cls._lock:
            best = cls.cache.get(key)            # another worker may have tuned this shape meanwhile
            if best is not None:
                return best
            # Candidates are timed on the caller's stream while other workers' kernels may share the device; that
            # can only cost speed (every candidate returns the same bits), and the min over repeats damps it.
            times = []
            for t in candidates:
                launch(t)
                best_t = float("inf")
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(2):
                        launch(t)
                    e1.record()
                    e1.synchronize()
                    best_t = min(best_t, e0.elapsed_time(e1) / 2)
                times.append(best_t)
            # candidates are listed in order of preference (simpler kernels first): a later one must win by more than the
            # timing noise of this short protocol (2 %) to be chosen — near-ties went either way from run to run and the
            # in-situ cost of the bigger tiles (one workgroup per CU, nothing else resident) is the higher one
            tmin = min(times)
            best = next(c for c, t in zip(candidates, times) if t <= 1.02 * tmin)
            if os.environ.get("FCP_AUTOTUNE_LOG"):
                print("autotune", key, {str(c): round(t * 1e3, 1) for c, t in zip(candidates, times)}, "->", best, flush=True)
            cls.cache[key] = best
            cls._disk_dirty = True
        if os.environ.get("FCP_TUNE_CACHE") != "0":
            cls._save_soon()
        return best

    _last_save = 0.0
    _atexit = False

    @classmethod
    def _save_soon(cls):
        """Persist new picks at most once every two seconds (a tuning pass meets dozens of new shapes in a row and every save is
        a read-modify-write of the user's file), and once more at interpreter exit if something is still unsaved."""
        import time
        if not cls._atexit:
            import atexit
            cls._atexit = True
            atexit.register(cls._save_quietly)
        if time.monotonic() - cls._last_save >= 2.0:
            cls._save_quietly()

    @classmethod
    def _save_quietly(cls):
        import time
        if not cls._disk_dirty or os.environ.get("FCP_TUNE_CACHE") == "0":
            return
        try:
            cls.save()
        except Exception:                                    # read-only home, full disk, a damaged file: tuning still works in-process
            pass
        cls._last_save = time.monotonic()


class ConvStats:
    """Algorithmic-FLOP accounting of conv launches (bench / roofline)."""
    enabled = False
    flops = 0
    launches = 0
    timing = None  # list of (start_event, end_event, flops, label, algorithmic bytes) when per-launch timing is on
    # list of (label, flops, algorithmic bytes, relaunch) when capture is on: ``relaunch()`` enqueues the very same launch again
    # (same descriptor, same tensors — kept alive by the closure) on the current stream.  tools/launch_ledger.py loops each
    # launch of a step on its own while clock / power are sampled (ctypes boundary only: the registered ops build no descriptor).
    replay = None

    @classmethod
    def note(cls, timing, e0, e1, flops, label, byts, relaunch=None):
        if timing is not None:
            e1.record()
            timing.append((e0, e1, flops, label, byts))
        if cls.replay is not None and relaunch is not None:
            cls.replay.append((label, flops, byts, relaunch))

    @classmethod
    def reset(cls):
        cls.flops, cls.launches = 0, 0
        if cls.timing is not None:
            cls.timing = []


class RangeMonitor:
    """Range guard of the fp16x3 path.  binary16 hi parts saturate at 65504 and the split silently loses its lo part
    below 2^-14; the generated weights keep activations O(1-100), a real checkpoint has never been through this code
    (no network in the build container).  While a monitor is active every conv / chain / stem launch is followed by one
    ``fcp_absmax_nhwc`` launch on its output view; ``report()`` returns [(label, max |x|)] in launch order."""
    active = None            # the monitor launches report to (one at a time: a diagnostic, not a data-path feature)
    LIMIT = 32768.0          # 2^15: one binade of headroom below binary16's largest finite value

    def __init__(self):
        self.rows = []       # (label, device scalar)

    def __enter__(self):
        assert RangeMonitor.active is None, "RangeMonitor is not re-entrant"
        RangeMonitor.active = self
        return self

    def __exit__(self, *exc):
        RangeMonitor.active = None

    def see(self, label: str, t: Act):
        m = torch.zeros((), dtype=torch.float32, device=t.buf.device)
        N.check(N.lib().fcp_absmax_nhwc(t.ptr(), t.n * t.h * t.w, t.c, t.ld, t.fmt, N.ptr(m), N.stream_ptr()), "fcp_absmax_nhwc")
        self.rows.append((label, m))

    def report(self):
        return [(label, float(m.item())) for label, m in self.rows]

    def check(self, what: str):
        rep = self.report()
        bad = [(label, v) for label, v in rep if not v < self.LIMIT]
        if bad:
            worst = max(bad, key=lambda r: r[1])
            raise FloatingPointError(
                f"{what}: activations leave the range of the fp16x3 (split binary16) conv path: |x| reaches {worst[1]:.4g} "
                f"after '{worst[0]}' ({len(bad)} of {len(rep)} launches at or above 2^15 = {self.LIMIT:.0f}; binary16 "
                f"saturates at 65504).  Load the model with precision='f32' (exact fp32 MFMA path) for these weights.")
        return rep


def selfcheck_mode(weights) -> bool:
    """Whether a model's ``load`` runs its range / accuracy self-check: FCP_SELFCHECK=1 always, 0 never; by default
    ("auto") whenever the weights come from a checkpoint — a file, the hub cache or a download — i.e. are not this
    package's own generated ones or a state dict the caller built in memory."""
    mode = os.environ.get("FCP_SELFCHECK", "auto")
    from_checkpoint = weights is None or (isinstance(weights, str) and weights != "generated")
    return mode == "1" or (mode == "auto" and from_checkpoint)


SELFCHECK_HARD_TOL = 1e-2      # fp16x3 vs exact fp32 beyond this fraction of the output's largest value: not a calibration question


def selfcheck_compare(what: str, got: torch.Tensor, ref: torch.Tensor, rel_tol: float) -> float:
    """max |got - ref| relative to the reference map's largest value (diagnostic arithmetic, not the data path).

    Three bands.  <= ``rel_tol``: silent.  (``rel_tol``, ``SELFCHECK_HARD_TOL``]: a WARNING — the tolerance was set on this
    package's generated weights, the release checkpoints have never been through this path (no network in the build
    container), and a calibration miss must not make the default ``Cropper()`` unusable; FCP_SELFCHECK=1 (the strict
    check) raises here too.  > ``SELFCHECK_HARD_TOL`` (or a non-finite difference): ``FloatingPointError`` in every mode —
    the two paths compute different functions of these weights; a model loaded with the default precision then falls back
    to the exact-fp32 path by itself (``selfcheck_at_load``).  The range guard (``RangeMonitor``: |x| >= 2^15 saturates
    binary16) is a hard error in every mode as well."""
    scale = float(ref.abs().max().item())
    diff = float((got - ref).abs().max().item()) / max(scale, 1e-30)
    if not diff <= rel_tol:
        msg = (f"{what}: the fp16x3 path and the exact-fp32 path disagree by {diff:.3g} of the output's "
               f"largest value (tolerance {rel_tol:g}, hard limit {SELFCHECK_HARD_TOL:g}) with these weights; load with precision='f32'.")
        if os.environ.get("FCP_SELFCHECK", "auto") == "1" or not diff <= SELFCHECK_HARD_TOL:
            raise FloatingPointError(msg)
        import warnings
        warnings.warn(msg)
    return diff


def selfcheck_at_load(model, sd, weights, precision, repack_f32):
    """The guard as ``load`` runs it.  Nothing to do on the exact-fp32 path or when ``selfcheck_mode`` says no check.  A
    check that fails — an activation past 2^15, or the two paths further apart than ``SELFCHECK_HARD_TOL`` — is

    * raised, when the caller asked for this precision by name (``precision="f16x3"``) or for the strict check
      (FCP_SELFCHECK=1): they get exactly what they asked for or an error;
    * otherwise (default precision) turned into a RuntimeWarning and an automatic reload on the exact-fp32 MFMA path
      (``repack_f32()``), which takes any checkpoint the reference takes (_layers.py:16-25) at ~1/3 of the speed; the
      report keeps the reason (``selfcheck_report["fallback"]``).
    """
    if model.precision != 1 or not selfcheck_mode(weights):
        return
    try:
        model.selfcheck(sd)
    except FloatingPointError as e:
        if precision is not None or os.environ.get("FCP_SELFCHECK", "auto") == "1":
            raise
        import warnings
        warnings.warn(f"{e}  Falling back to precision='f32' for this model.", RuntimeWarning)
        repack_f32()
        model.selfcheck_report = {"fallback": "f32", "reason": str(e)}


def _monitor(label, *outs):
    mon = RangeMonitor.active
    if mon is not None:
        for i, t in enumerate(outs):
            if t is not None and t.c % 8 == 0:
                mon.see(label if i == 0 else f"{label} [output {i}]", t)


def conv(pc: PackedConv, x: Act, out: Act | None = None, *, act_slope: float = 1.0,
         alpha: float = 1.0, res1: Act | None = None, res1_pre: bool = True,
         res2: Act | None = None, alpha2: float = 1.0, in_up2: bool = False,
         tile_n: int | None = None, out_fmt: int = 0, tile_m: int | None = None,
         x2: Act | None = None, x2_stride: int = 1, flat: bool = False, balance_tail: bool = False,
         band: tuple[int, int] = (0, 0)) -> Act:
    """Launch one fused convolution.  ``act_slope``: 1 = identity, 0 = ReLU.  ``out_fmt`` selects the
    format of a freshly allocated output (an explicit ``out`` view carries its own).  ``x2``: second
    source of a 1x1 conv — the filter's trailing ``x2.c`` input channels read ``x2`` at
    ``(ho*x2_stride, wo*x2_stride)`` (both sources split32, fp16x3 path).  ``flat``: force the 64-bit
    flat-addressing variant of the fp32 kernel (what tensors >= 4 GiB take on their own).  ``balance_tail`` (with an
    explicit ``tile_m=256``): the balanced M-tile schedule of the 256-row kernel (the autotuner picks it by itself).
    ``band`` = (rows above, rows below): real rows the input view holds around the rows the output view covers, in place of the
    zero padding at the view's edge (``fcp_conv_desc.band_top / band_bottom``): rows [a, b) of a larger image's conv, exactly."""
    assert x.c + (x2.c if x2 is not None else 0) == pc.cin, f"conv expects {pc.cin} input channels, got {x.c}"
    in_h, in_w = (x.h * 2, x.w * 2) if in_up2 else (x.h, x.w)
    oh = (in_h - band[0] - band[1] + 2 * pc.pad - pc.kh) // pc.stride + 1
    ow = (in_w + 2 * pc.pad - pc.kw) // pc.stride + 1
    if out is None:
        out = Act.empty(x.n, oh, ow, pc.cout, x.buf.device, out_fmt)
    assert (out.n, out.h, out.w, out.c) == (x.n, oh, ow, pc.cout), "conv: bad output view"
    if (x.fmt or out.fmt or (res1 is not None and res1.fmt) or (res2 is not None and res2.fmt)) and pc.precision != 1:
        raise ValueError("split32 tensors can only be used with filters packed for the fp16x3 path")
    m = x.n * oh * ow
    d = N.ConvDesc()
    d.in_, d.w, d.out = x.ptr(), N.ptr(pc.w), out.ptr()
    if x2 is not None:
        assert x2.fmt == 1 and x.fmt == 1 and x2.n == x.n, "two-source convs take split32 tensors"
        d.in2, d.cin2, d.in2_ld, d.in2_h, d.in2_w, d.in2_stride = x2.ptr(), x2.c, x2.ld, x2.h, x2.w, x2_stride
    d.bias = N.ptr(pc.bias)
    d.wscale = N.ptr(pc.wscale)
    d.precision = pc.precision
    d.in_fmt, d.out_fmt = x.fmt, out.fmt
    d.res1_fmt = res1.fmt if res1 is not None else 0
    d.res2_fmt = res2.fmt if res2 is not None else 0
    d.res1 = res1.ptr() if res1 is not None else None
    d.res2 = res2.ptr() if res2 is not None else None
    d.n, d.in_h, d.in_w = x.n, in_h, in_w
    d.cin, d.in_ld, d.in_up2 = pc.cin, x.ld, int(in_up2)
    d.cout, d.kh, d.kw, d.stride, d.pad = pc.cout, pc.kh, pc.kw, pc.stride, pc.pad
    d.out_h, d.out_w, d.out_ld = oh, ow, out.ld
    d.tile_n = tile_n or _pick_tile_n(pc.cout, m)
    d.tile_m = tile_m or 128
    d.cin4 = int(pc.cin4)
    d.act_slope, d.alpha, d.alpha2 = act_slope, alpha, alpha2
    d.res1_pre = int(res1_pre)
    d.flags = (N.CONV_FLAT_ADDR if flat else 0) | (N.CONV_BALANCE_TAIL if balance_tail else 0)
    d.cu_budget = getattr(_budget, "cus", 0)
    d.band_top, d.band_bottom = band
    if res1 is not None:
        assert res1.c == pc.cout and res1.n == x.n
        d.res1_ld, d.res1_h, d.res1_w = res1.ld, res1.h, res1.w
    if res2 is not None:
        assert (res2.n, res2.h, res2.w, res2.c) == (x.n, oh, ow, pc.cout)
        d.res2_ld = res2.ld
    big_ok = (pc.precision == 1 and x.fmt == 1 and not pc.cin4 and not in_up2 and pc.cout % 8 == 0
              and pc.cout >= 128 and m >= 256 * 64)
    halo_ok = (pc.precision == 1 and x.fmt == 1 and not pc.cin4 and (pc.kh, pc.kw, pc.stride, pc.pad) == (3, 3, 1, 1)
               and pc.cout <= 64 and pc.cout % 8 == 0 and pc.cin >= 64 and x2 is None and (res1 is None or (res1.h, res1.w) == (oh, ow)))
    # the wide halo-tile kernel (column tiles inner, filters through a tap ring): cin % 64 == 0, 65 .. 128 filters, no residuals
    # (it also runs 33 .. 64 filters, tile (1, 64): 586 vs 600 us alone on RRDB's conv5, nothing end to end: not offered)
    wide_ok = (pc.precision == 1 and x.fmt == 1 and not pc.cin4 and (pc.kh, pc.kw, pc.stride, pc.pad) == (3, 3, 1, 1)
               and pc.cout % 8 == 0 and pc.cin % 64 == 0 and x2 is None and 64 < pc.cout <= 128 and res1 is None and res2 is None)
    if tile_n is None and tile_m is None and (pc.cout > 64 or halo_ok):
        Autotune.ensure_loaded()
    if tile_n is None and tile_m is None and (pc.cout > 64 or halo_ok) and (Autotune.enabled or Autotune.cache):
        key = (pc.cin, pc.cout, pc.kh, pc.kw, pc.stride, m, int(in_up2), res1 is not None, res2 is not None,
               pc.precision, x.fmt, out.fmt, None if x2 is None else (x2.c, x2_stride), d.cu_budget)
        cands = tile_candidates(pc.cout, halo_ok, wide_ok and HALO_WIDE, big_ok and BIG_TILES, BALANCE_TAIL)
        best = Autotune.cache.get(key)          # a tuned shape keeps its tile after tuning is switched off
        if best is not None and tuple(best) not in cands:
            with Autotune._lock:                # (save() walks the cache under the same lock in another GPU worker thread)
                Autotune.cache.pop(key, None)   # a pick from an older table that this build no longer offers
            best = None
        if best is None and Autotune.enabled:
            # Tuning launches the op several times.  An op whose output aliases one of its inputs (RRDB's last dense-block
            # conv writes the buffer its second residual is read from) is not idempotent: its trial launches write a
            # scratch tensor of the same geometry instead, and only the final launch below touches the real output.
            # Only a real overlap counts: reading and writing disjoint channel slices of one buffer (RRDB's dense-block
            # convs, SSH's concat) is idempotent.  The scratch holds just the output view (its own pixel pitch), not a
            # copy of the whole buffer.
            real_out, real_ld = d.out, d.out_ld
            same = lambda t: t is not None and t.buf.untyped_storage().data_ptr() == out.buf.untyped_storage().data_ptr()
            aliased = any(same(t) and t.c0 < out.c0 + out.c and out.c0 < t.c0 + t.c for t in (x, x2, res1, res2))
            scratch = torch.empty((out.n, out.h, out.w, out.c), dtype=torch.float32, device=out.buf.device) if aliased else None

            base_flags = d.flags

            def _launch(t):
                d.tile_m, d.tile_n = t[0], t[1]
                d.flags = base_flags | (t[2] if len(t) > 2 else 0)
                if scratch is not None:
                    d.out, d.out_ld = N.ptr(scratch), out.c
                N.check(N.lib().fcp_conv2d_nhwc_f32(C.byref(d), N.stream_ptr()), "fcp_conv2d_nhwc_f32")
                d.out, d.out_ld = real_out, real_ld
            best = Autotune.pick(key, cands, _launch)
            d.flags = base_flags
        if best is not None:
            d.tile_m, d.tile_n = best[0], best[1]
            d.flags |= best[2] if len(best) > 2 else 0
    timing = ConvStats.timing
    if timing is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    if T.ENABLED:            # FCP_BOUNDARY=torch: the same launch through the registered PyTorch custom op (out variant)
        T.load().conv2d_out(x.buf, x.c0, pc.cin, pc.w, pc.bias, pc.wscale, None if res1 is None else res1.buf,
                            0 if res1 is None else res1.c0, None if res2 is None else res2.buf, 0 if res2 is None else res2.c0,
                            out.buf, out.c0, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad, float(act_slope), float(alpha), float(alpha2),
                            bool(res1_pre), pc.precision, x.fmt, out.fmt, d.res1_fmt, d.res2_fmt, bool(in_up2), bool(pc.cin4),
                            int(d.tile_m), int(d.tile_n), None if x2 is None else x2.buf, 0 if x2 is None else x2.c0,
                            0 if x2 is None else x2.c, int(x2_stride), int(d.flags), int(d.cu_budget), int(band[0]), int(band[1]))
    else:
        N.check(N.lib().fcp_conv2d_nhwc_f32(C.byref(d), N.stream_ptr()), "fcp_conv2d_nhwc_f32")
    if timing is not None or ConvStats.replay is not None:
        byts = 4 * (m * (pc.cout + (pc.cout if res1 is not None else 0) + (pc.cout if res2 is not None else 0))
                    + x.n * in_h * in_w * x.c // (4 if in_up2 else 1) + (m * x2.c if x2 is not None else 0)
                    + pc.cout * pc.cin * pc.kh * pc.kw)
        keep = (pc, x, x2, res1, res2, out)
        ConvStats.note(timing, e0 if timing is not None else None, e1 if timing is not None else None, pc.flops_per_pixel * m,
                       f"conv {pc.kh}x{pc.kw} s{pc.stride} {pc.cin}->{pc.cout} @{oh}x{ow} tile {d.tile_m}x{d.tile_n}"
                       f"{' bal' if d.flags & N.CONV_BALANCE_TAIL else ''}{' +res' if res1 is not None else ''}", byts,
                       lambda d=d, keep=keep: N.check(N.lib().fcp_conv2d_nhwc_f32(C.byref(d), N.stream_ptr()), "fcp_conv2d_nhwc_f32"))
    if ConvStats.enabled:
        ConvStats.flops += pc.flops_per_pixel * m
        ConvStats.launches += 1
    if RangeMonitor.active is not None:
        _monitor(f"conv {pc.kh}x{pc.kw} s{pc.stride} {pc.cin}->{pc.cout} @{oh}x{ow}", out)
    return out


def tile_candidates(cout: int, halo_ok: bool, wide_ok: bool, big_ok: bool, balance_tail: bool = True):
    """The tile vocabulary the tuner chooses from for one conv shape, in order of preference (``Autotune.pick``).  The tuned
    tables store these tuples: a change of their MEANING (not the mere addition of a shape class) needs a new
    ``Autotune.TABLE_VERSION`` — tests/test_autotune_cache_cpu.py pins the vocabulary's digest to the version."""
    cands = [(128, 128), (128, 64)]
    if halo_ok:
        cands = ([(128, 32)] if cout <= 32 else []) + [(128, 64), (1, 32)]
    if wide_ok:
        cands = cands + [(1, 128)]
    if big_ok:
        big = [(256, 128)] + ([(256, 256)] if cout >= 256 else []) + ([(256, 192)] if 128 < cout <= 192 else [])
        cands += big
        if balance_tail:          # the same tiles with the last dispatch round cut into shorter M-tiles (same bits)
            cands += [(tm, tn, N.CONV_BALANCE_TAIL) for tm, tn in big]
    return cands


def chain_supported(pc2: PackedConv | None, pc3: PackedConv, pc1n: PackedConv | None, residual: bool = True, cb: int = 0) -> bool:
    """Shapes ``fcp_bottleneck_chain_f16x3`` covers, all packed for the fp16x3 path with folded-BN bias:
    * with conv2: 64-wide bottleneck (3x3 64->64 / 1, 1x1 64->256 + residual), next conv1 1x1 256 -> 64 | 128;
    * pair (``pc2`` None): 1x1 128->512 + residual, next conv1 512->128 | 256  (layer-2 identity blocks; the last one
      with layer3.0.conv1),
      1x1 256->1024 + residual, next conv1 1024->256  (layer-3 identity blocks), or
      1x1 128->256 without residual, next conv1 256->64  (layer1.0's conv3 + downsample K-concat, layer1.1.conv1);
    * two-source pair (``cb`` = channels of the second source): 1x1 (128 + 256)->512 without residual, next conv1 512->128
      (layer2.0's conv3 + stride-2 downsample over [conv2 out | x(::2, ::2)], layer2.1.conv1);
    * expand form (``pc1n`` None): 1x1 256->1024 + residual alone — conv3 + identity of a layer-3 block with its operand
      fragments in registers and two workgroups per CU; the next conv1 is then an ordinary launch."""
    convs = [pc for pc in (pc2, pc3, pc1n) if pc is not None]
    if not all(pc.precision == 1 and pc.bias is not None and not pc.cin4 for pc in convs):
        return False
    one = lambda pc, cin, cout: (pc.cin, pc.cout, pc.kh, pc.kw, pc.stride, pc.pad) == (cin, cout, 1, 1, 1, 0)
    if pc1n is None:          # expand form: conv3 + identity only (1x1 256 -> 1024 + residual; layer 3), no next conv1
        return pc2 is None and residual and not cb and one(pc3, 256, 1024)
    if cb:
        return pc2 is None and not residual and cb == 256 and one(pc3, 384, 512) and one(pc1n, 512, 128) and CHAIN_TWO_SOURCE
    if pc2 is not None:
        return ((pc2.cin, pc2.cout, pc2.kh, pc2.kw, pc2.stride, pc2.pad) == (64, 64, 3, 3, 1, 1) and residual
                and one(pc3, 64, 256) and (one(pc1n, 256, 64) or one(pc1n, 256, 128)))
    if residual:
        return ((one(pc3, 128, 512) and (one(pc1n, 512, 128) or one(pc1n, 512, 256)))
                or (one(pc3, 256, 1024) and one(pc1n, 1024, 256)))
    return one(pc3, 128, 256) and one(pc1n, 256, 64)


HALO_WIDE = os.environ.get("FCP_HALO_WIDE", "1") != "0"       # A/B switch: offer the wide halo-tile kernel to the tile tuner
CHAIN_TILE_M = int(os.environ.get("FCP_CHAIN_TILE_M", "0"))   # 0 / 128: 4-wave tiles of 128 pixels; 256: 8-wave tiles where they fit
# conv2 forms (layer 1): 8 x 16 pixel patches whose halo is staged once per channel slice (tile_m = 16 in the descriptor)
# instead of 128 consecutive pixels fetched once per tap; same bits.  FCP_CHAIN_PATCH=0 = the linear tiles.
CHAIN_PATCH = os.environ.get("FCP_CHAIN_PATCH", "1") != "0"
# ... and a block whose output only a stride-2 consumer reads stores the even pixels only (bottleneck_chain(out_even_only=True))
CHAIN_SPARSE_OUT = os.environ.get("FCP_CHAIN_SPARSE_OUT", "1") != "0"
# layer2.0's two-source conv3 (+ downsample) and layer2.1.conv1 as one pair launch (A/B switch: 0 = the two conv launches)
CHAIN_TWO_SOURCE = os.environ.get("FCP_CHAIN_TWO_SOURCE", "1") != "0"
# layer-3 identity blocks: "pair-only" (default) = conv3 + identity + next conv1 in one launch (one wave per SIMD), the stand-alone
# layer3.5.conv3 on the tuned conv tile; "pair" = the same, layer3.5.conv3 on the expand form; "expand" = every conv3 + identity on
# the expand form (two workgroups per CU) followed by an ordinary conv1 launch.  Same bits; a three-way tie in the in-call A/B
# (profiles/r05_probes.md section 10), so the product keeps the launch mix its profiles were taken with.
L3_FORM = os.environ.get("FCP_L3_FORM", "pair-only")


def bottleneck_chain(pc2: PackedConv | None, pc3: PackedConv, pc1n: PackedConv, t1: Act, res: Act | None,
                     out: Act | None = None, t1n: Act | None = None, tile_m: int | None = None, out_even_only: bool = False,
                     t1b: Act | None = None, t1b_stride: int = 1):
    """One launch for  out = relu(conv3(relu(conv2(t1))) [+ res]),  t1n = relu(conv1n(out))  (BatchNorm folded): conv2 /
    conv3 of a bottleneck and conv1 of the next block; ``pc2`` None: the pair forms (no conv2, see ``chain_supported``).
    Bit-identical to the separate ``conv`` calls.  Returns (out, t1n), both split32.  ``out_even_only`` (conv2 forms on patch
    tiles): ``out`` is only stored at pixels with even y and even x — for a block whose output nothing but a stride-2 consumer
    reads (the other three quarters of the tensor are never written nor read; ``t1n`` is complete).  Ignored, i.e. a full
    ``out``, where the patch form is not in use or a ``RangeMonitor`` wants to see the whole tensor.  ``t1b`` (two-source
    pair): the trailing ``t1b.c`` input channels of conv3 are read from ``t1b`` at ``(y * t1b_stride, x * t1b_stride)`` —
    what ``conv(..., x2=, x2_stride=)`` does for the stand-alone two-source conv."""
    cb = t1b.c if t1b is not None else 0
    assert chain_supported(pc2, pc3, pc1n, res is not None, cb), "bottleneck_chain: unsupported shapes"
    if pc1n is None:
        return _expand_conv3(pc3, t1, res, out)
    assert t1.fmt == 1 and t1.c + cb == pc3.cin and (res is None or (res.fmt == 1 and res.c == pc3.cout))
    assert t1b is None or (t1b.fmt == 1 and t1b.n == t1.n and (t1.h - 1) * t1b_stride < t1b.h and (t1.w - 1) * t1b_stride < t1b.w)
    assert res is None or (t1.n, t1.h, t1.w) == (res.n, res.h, res.w)
    dev = t1.buf.device
    m = t1.n * t1.h * t1.w
    flops = ((pc2.flops_per_pixel if pc2 is not None else 0) + pc3.flops_per_pixel + pc1n.flops_per_pixel) * m
    timing = ConvStats.timing
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    opt = lambda pc, f: None if pc is None else getattr(pc, f)
    if tile_m is None:
        tile_m = 16 if (pc2 is not None and CHAIN_PATCH and CHAIN_TILE_M == 0) else (CHAIN_TILE_M if (pc2 is not None or CHAIN_TILE_M not in (16, 32)) else 0)
    flags = N.CHAIN_OUT_EVEN_ONLY if (out_even_only and tile_m in (16, 32) and CHAIN_SPARSE_OUT and RangeMonitor.active is None) else 0
    d = None
    if T.ENABLED and out is None and t1n is None:
        # FCP_BOUNDARY=torch: the registered custom op allocates and returns both tensors
        o, t = T.load().bottleneck_chain(t1.buf, t1.c0, None if res is None else res.buf, 0 if res is None else res.c0,
                                         opt(pc2, "w"), opt(pc2, "wscale"), opt(pc2, "bias"), pc3.w, pc3.wscale, pc3.bias,
                                         pc1n.w, pc1n.wscale, pc1n.bias, pc3.cin, pc3.cout, pc1n.cout, tile_m, flags,
                                         None if t1b is None else t1b.buf, 0 if t1b is None else t1b.c0, cb, int(t1b_stride))
        out, t1n = Act(o, fmt=1), Act(t, fmt=1)
    else:
        if out is None:
            out = Act.empty(t1.n, t1.h, t1.w, pc3.cout, dev, 1)
        if t1n is None:
            t1n = Act.empty(t1.n, t1.h, t1.w, pc1n.cout, dev, 1)
        assert out.fmt == 1 and t1n.fmt == 1 and (out.n, out.h, out.w, out.c) == (t1.n, t1.h, t1.w, pc3.cout)
        assert (t1n.n, t1n.h, t1n.w, t1n.c) == (t1.n, t1.h, t1.w, pc1n.cout)
        d = N.ChainDesc()
        d.t1, d.res, d.out, d.t1n = t1.ptr(), (res.ptr() if res is not None else None), out.ptr(), t1n.ptr()
        d.w2, d.ws2, d.b2 = N.ptr(opt(pc2, "w")), N.ptr(opt(pc2, "wscale")), N.ptr(opt(pc2, "bias"))
        d.w3, d.ws3, d.b3 = N.ptr(pc3.w), N.ptr(pc3.wscale), N.ptr(pc3.bias)
        d.w1n, d.ws1n, d.b1n = N.ptr(pc1n.w), N.ptr(pc1n.wscale), N.ptr(pc1n.bias)
        d.n, d.h, d.w, d.c, d.cn, d.nout = t1.n, t1.h, t1.w, pc3.cin, pc1n.cout, pc3.cout    # c: all of conv3's input channels
        d.t1_ld, d.res_ld, d.out_ld, d.t1n_ld = t1.ld, (res.ld if res is not None else 0), out.ld, t1n.ld
        d.tile_m, d.flags = tile_m, flags
        if t1b is not None:
            d.t1b, d.cb, d.t1b_ld, d.t1b_h, d.t1b_w, d.t1b_stride = t1b.ptr(), cb, t1b.ld, t1b.h, t1b.w, t1b_stride
        N.check(N.lib().fcp_bottleneck_chain_f16x3(C.byref(d), N.stream_ptr()), "fcp_bottleneck_chain_f16x3")
    if timing is not None or ConvStats.replay is not None:
        c, nout, cn = pc3.cin, pc3.cout, pc1n.cout
        byts = 4 * (m * (c + (nout if not flags else nout // 4) + (nout if res is not None else 0) + cn) + (0 if pc2 is None else 9 * c * c)
                    + c * nout + nout * cn)
        relaunch = None
        if d is not None:
            keep = (pc2, pc3, pc1n, t1, res, out, t1n, t1b)
            relaunch = lambda d=d, keep=keep: N.check(N.lib().fcp_bottleneck_chain_f16x3(C.byref(d), N.stream_ptr()), "fcp_bottleneck_chain_f16x3")
        ConvStats.note(timing, e0 if timing is not None else None, e1 if timing is not None else None, flops,
                       f"chain {'3x3 ' if pc2 is not None else ''}{c}->{nout}->{cn} @{t1.h}x{t1.w}"
                       f"{' +res' if res is not None else ''}{' out@even' if flags else ''}{' two-source' if t1b is not None else ''}", byts, relaunch)
    if ConvStats.enabled:
        ConvStats.flops += flops
        ConvStats.launches += 1
    if RangeMonitor.active is not None:
        _monitor(f"chain {'3x3 ' if pc2 is not None else ''}{pc3.cin}->{pc3.cout}->{pc1n.cout} @{t1.h}x{t1.w}", out, t1n)
    return out, t1n


def _expand_conv3(pc3: PackedConv, t1: Act, res: Act, out: Act | None):
    """The expand form of ``fcp_bottleneck_chain_f16x3`` (cn = 0, no conv1' filter): out = relu(conv3(t1) + res), bit-identical to
    ``conv(pc3, t1, act_slope=0, res1=res, res1_pre=True, out_fmt=1)``.  Returns (out, None)."""
    assert t1.fmt == 1 and res is not None and res.fmt == 1 and (t1.n, t1.h, t1.w) == (res.n, res.h, res.w)
    m = t1.n * t1.h * t1.w
    via_op = T.ENABLED and out is None        # the registered op allocates its output; an explicit view goes through the C ABI
    if out is None and not via_op:
        out = Act.empty(t1.n, t1.h, t1.w, pc3.cout, t1.buf.device, 1)
    assert out is None or (out.fmt == 1 and (out.n, out.h, out.w, out.c) == (t1.n, t1.h, t1.w, pc3.cout))
    timing = ConvStats.timing
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if via_op:
        # FCP_BOUNDARY=torch: the same registered op as the other chain forms (no next-conv1 filter, cn = 0)
        o, _ = T.load().bottleneck_chain(t1.buf, t1.c0, res.buf, res.c0, None, None, None, pc3.w, pc3.wscale, pc3.bias,
                                         None, None, None, pc3.cin, pc3.cout, 0, 0, 0, None, 0, 0, 1)
        out = Act(o, fmt=1)
    else:
        d = N.ChainDesc()
        d.t1, d.res, d.out = t1.ptr(), res.ptr(), out.ptr()
        d.w3, d.ws3, d.b3 = N.ptr(pc3.w), N.ptr(pc3.wscale), N.ptr(pc3.bias)
        d.n, d.h, d.w, d.c, d.cn, d.nout = t1.n, t1.h, t1.w, pc3.cin, 0, pc3.cout
        d.t1_ld, d.res_ld, d.out_ld = t1.ld, res.ld, out.ld
        N.check(N.lib().fcp_bottleneck_chain_f16x3(C.byref(d), N.stream_ptr()), "fcp_bottleneck_chain_f16x3")
    flops = pc3.flops_per_pixel * m
    if timing is not None or ConvStats.replay is not None:
        relaunch = None
        if not via_op:
            keep = (pc3, t1, res, out)
            relaunch = lambda d=d, keep=keep: N.check(N.lib().fcp_bottleneck_chain_f16x3(C.byref(d), N.stream_ptr()), "fcp_bottleneck_chain_f16x3")
        ConvStats.note(timing, e0 if timing is not None else None, e1 if timing is not None else None, flops,
                       f"expand {pc3.cin}->{pc3.cout} @{t1.h}x{t1.w} +res", 4 * (m * (pc3.cin + 2 * pc3.cout) + pc3.cin * pc3.cout), relaunch)
    if ConvStats.enabled:
        ConvStats.flops += flops
        ConvStats.launches += 1
    if RangeMonitor.active is not None:
        _monitor(f"expand {pc3.cin}->{pc3.cout} @{t1.h}x{t1.w}", out)
    return out, None


def u8_to_nhwc4(images_u8: torch.Tensor, sub=(0.0, 0.0, 0.0), div: float = 1.0) -> Act:
    """(n,h,w,3) uint8 device tensor -> fp32 NHWC4 activation ((x - sub) / div)."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[3] == 3
    assert images_u8.is_contiguous()
    n, h, w, _ = images_u8.shape
    out = Act.empty(n, h, w, 4, images_u8.device)
    sub_arr = (C.c_float * 3)(*[float(s) for s in sub])
    N.check(N.lib().fcp_u8_to_nhwc4_f32(N.ptr(images_u8), out.ptr(), n * h * w, sub_arr, float(div),
                                        N.stream_ptr()), "fcp_u8_to_nhwc4_f32")
    return out


@dataclass
class PackedStem:
    wfrag: torch.Tensor      # int16 [2][11][2][64][8]: binary16 hi / lo filter fragments (MFMA operand order)
    bias: torch.Tensor
    wscale: torch.Tensor
    flops_per_pixel: int     # algorithmic FLOP per stem output pixel (2 * 64 * 3 * 49)


def pack_stem_fused(weight, bn, device, cin_perm=None) -> PackedStem:
    """Filter of the fused uint8 -> 7x7/2 conv -> ReLU -> max-pool kernel (fcp_stem7x7s2_relu_pool_u8).
    weight (64,3,7,7); K index = kh*24 + kw*3 + c, 22 chunks of 8 (the last one, and k % 24 >= 21, zero)."""
    if isinstance(weight, torch.Tensor):
        weight = weight.detach().cpu().numpy()
    bn = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in bn.items()}
    w, b = fold_bn(weight, bn, None)
    if cin_perm is not None:
        w = w[:, cin_perm]
    assert w.shape == (64, 3, 7, 7), "the fused stem is the ResNet 7x7 / 64-filter stem"
    wk = np.zeros((64, 22 * 8), np.float32)
    wk.reshape(64, 22, 8)                                        # (filter, chunk, element) view of the same K axis
    for kh in range(7):
        wk[:, kh * 24:kh * 24 + 21] = w[:, :, kh, :].transpose(0, 2, 1).reshape(64, 21)     # (kw, c) fastest
    amax = np.abs(wk).max(1)
    e = np.where(amax > 0, np.floor(np.log2(np.maximum(amax, 1e-38))), 0.0)
    scale = np.exp2(e).astype(np.float32)
    ws = wk / scale[:, None]
    hi = ws.astype(np.float16)
    lo = (ws - hi.astype(np.float32)).astype(np.float16)
    frag = np.zeros((2, 11, 2, 64, 8), np.float16)
    lane = np.arange(64)
    for ct in range(2):
        n = ct * 32 + (lane & 31)
        for q in range(11):
            ch = 2 * q + (lane >> 5)
            k = ch[:, None] * 8 + np.arange(8)[None, :]
            frag[ct, q, 0] = hi[n[:, None], k]
            frag[ct, q, 1] = lo[n[:, None], k]
    return PackedStem(torch.from_numpy(np.ascontiguousarray(frag.view(np.int16))).to(device),
                      torch.from_numpy(np.ascontiguousarray(b)).to(device), torch.from_numpy(scale).to(device),
                      2 * 64 * 3 * 49)


def stem_conv1_supported(pc1: PackedConv | None) -> bool:
    """conv1 of the first bottleneck can ride in the stem launch: 1x1 / 1, 64 -> 64, fp16x3, folded-BN bias."""
    return (pc1 is not None and pc1.precision == 1 and pc1.bias is not None and not pc1.cin4
            and (pc1.cin, pc1.cout, pc1.kh, pc1.kw, pc1.stride, pc1.pad) == (64, 64, 1, 1, 1, 0))


def stem_relu_pool_u8(ps: PackedStem, images_u8: torch.Tensor, out: Act | None = None, mean_rgb=(123, 117, 104),
                      out_fmt: int = 1, conv1: PackedConv | None = None, t1: Act | None = None):
    """(n,h,w,3) uint8 -> stem conv + ReLU + max-pool, one launch; ``out`` may be a 64-channel slice.
    ``conv1`` (see ``stem_conv1_supported``): the same launch also computes ``t1 = relu(conv1(pooled))`` (split32) and
    returns ``(out, t1)`` — bit-identical to ``conv(conv1, out, act_slope=0.0, out_fmt=1)``."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[3] == 3 and images_u8.is_contiguous()
    n, h, w, _ = images_u8.shape
    hs, ws = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (hs - 1) // 2 + 1, (ws - 1) // 2 + 1
    if out is None:
        out = Act.empty(n, hp, wp, 64, images_u8.device, out_fmt)
    assert (out.n, out.h, out.w, out.c) == (n, hp, wp, 64), "stem: bad output view"
    mean = (C.c_int32 * 3)(*[int(m) for m in mean_rgb])
    timing = ConvStats.timing
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    flops = ps.flops_per_pixel * n * hs * ws
    if conv1 is not None:
        assert stem_conv1_supported(conv1) and out.fmt == 1, "stem + conv1: 1x1 64 -> 64 fp16x3 conv on a split32 pooled map"
        if t1 is None:
            t1 = Act.empty(n, hp, wp, 64, images_u8.device, 1)
        assert t1.fmt == 1 and (t1.n, t1.h, t1.w, t1.c) == (n, hp, wp, 64)
        flops += conv1.flops_per_pixel * n * hp * wp
        N.check(N.lib().fcp_stem7x7s2_relu_pool_conv1_u8(N.ptr(images_u8), n, h, w, mean, N.ptr(ps.wfrag), N.ptr(ps.bias),
                                                         N.ptr(ps.wscale), out.ptr(), out.ld, out.fmt, N.ptr(conv1.w),
                                                         N.ptr(conv1.wscale), N.ptr(conv1.bias), t1.ptr(), t1.ld, N.stream_ptr()),
                "fcp_stem7x7s2_relu_pool_conv1_u8")
    else:
        N.check(N.lib().fcp_stem7x7s2_relu_pool_u8(N.ptr(images_u8), n, h, w, mean, N.ptr(ps.wfrag), N.ptr(ps.bias),
                                                   N.ptr(ps.wscale), out.ptr(), out.ld, out.fmt, N.stream_ptr()),
                "fcp_stem7x7s2_relu_pool_u8")
    if timing is not None or ConvStats.replay is not None:
        byts = n * h * w * 3 + 4 * n * hp * wp * 64 * (2 if conv1 is not None else 1)
        ConvStats.note(timing, e0 if timing is not None else None, e1 if timing is not None else None, flops,
                       f"stem 7x7 s2 + pool{' + conv1' if conv1 is not None else ''} @{hp}x{wp}", byts,
                       lambda: stem_relu_pool_u8(ps, images_u8, out, mean_rgb, out_fmt, conv1, t1))   # (replayed with capture switched off)
    if ConvStats.enabled:
        ConvStats.flops += flops
        ConvStats.launches += 1
    if RangeMonitor.active is not None:
        _monitor(f"stem 7x7 s2 + pool @{hp}x{wp}", out, t1)
    return out if conv1 is None else (out, t1)


def stem_relu_pool_f32(ps: PackedStem, x4: Act, out: Act | None = None, out_fmt: int = 1) -> Act:
    """fp32 NHWC4 (n,h,w,4), already normalised -> 7x7 / 2 stem conv (BatchNorm folded) + ReLU + 3x3 / 2 max-pool in one
    launch (``fcp_stem7x7s2_relu_pool_f32``: BiSeNet's ResNet-18 stem); ``out`` may be a 64-channel slice.  The activation
    is split hi + lo while it is staged and a k-step is the full three-term product; K order (kh, kw, c), i.e. the
    accuracy class of ``conv`` + ``maxpool3x3s2`` with a different summation order."""
    assert x4.fmt == 0 and x4.c0 == 0 and x4.c == 4 and x4.ld == 4, "the fused fp32 stem reads a dense NHWC4 tensor"
    n, h, w = x4.n, x4.h, x4.w
    hs, ws = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (hs - 1) // 2 + 1, (ws - 1) // 2 + 1
    if out is None:
        out = Act.empty(n, hp, wp, 64, x4.buf.device, out_fmt)
    assert (out.n, out.h, out.w, out.c) == (n, hp, wp, 64), "stem: bad output view"
    timing = ConvStats.timing
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    N.check(N.lib().fcp_stem7x7s2_relu_pool_f32(x4.ptr(), n, h, w, N.ptr(ps.wfrag), N.ptr(ps.bias), N.ptr(ps.wscale),
                                                out.ptr(), out.ld, out.fmt, N.stream_ptr()), "fcp_stem7x7s2_relu_pool_f32")
    flops = ps.flops_per_pixel * n * hs * ws
    if timing is not None:
        ConvStats.note(timing, e0, e1, flops, f"stem(f32) 7x7 s2 + pool @{hp}x{wp}", n * h * w * 16 + 4 * n * hp * wp * 64)
    if ConvStats.enabled:
        ConvStats.flops += flops
        ConvStats.launches += 1
    if RangeMonitor.active is not None:
        _monitor(f"stem(f32) 7x7 s2 + pool @{hp}x{wp}", out)
    return out


def f32nchw_to_nhwc4(images: torch.Tensor, sub=(0.0, 0.0, 0.0), div: float = 1.0) -> Act:
    """(n,3,h,w) fp32 device tensor -> fp32 NHWC4 activation ((x - sub) / div)."""
    assert images.dtype == torch.float32 and images.dim() == 4 and images.shape[1] == 3
    images = images.contiguous()
    n, _, h, w = images.shape
    out = Act.empty(n, h, w, 4, images.device)
    sub_arr = (C.c_float * 3)(*[float(s) for s in sub])
    N.check(N.lib().fcp_f32nchw_to_nhwc4_f32(N.ptr(images), out.ptr(), n, h, w, sub_arr, float(div),
                                             N.stream_ptr()), "fcp_f32nchw_to_nhwc4_f32")
    return out


def maxpool3x3s2(x: Act, out: Act | None = None) -> Act:
    """``out`` may be a channel slice of a wider buffer (same format as ``x``)."""
    assert x.c0 == 0 and x.c == x.ld
    oh, ow = (x.h + 2 - 3) // 2 + 1, (x.w + 2 - 3) // 2 + 1
    if out is None:
        out = Act.empty(x.n, oh, ow, x.c, x.buf.device, x.fmt)
    assert (out.n, out.h, out.w, out.c, out.fmt) == (x.n, oh, ow, x.c, x.fmt), "maxpool: bad output view"
    fn = N.lib().fcp_maxpool3x3s2_split32 if x.fmt == 1 else N.lib().fcp_maxpool3x3s2_nhwc_f32
    N.check(fn(x.ptr(), out.ptr(), x.n, x.h, x.w, x.c, out.ld, oh, ow, N.stream_ptr()), "fcp_maxpool3x3s2")
    return out


def f32_to_split32(x: Act) -> Act:
    assert x.fmt == 0 and x.c0 == 0 and x.c == x.ld and x.c % 32 == 0
    out = Act.empty(x.n, x.h, x.w, x.c, x.buf.device, 1)
    N.check(N.lib().fcp_f32_to_split32(x.ptr(), out.ptr(), x.n * x.h * x.w, x.c, N.stream_ptr()), "fcp_f32_to_split32")
    return out


def split32_to_f32(x: Act) -> Act:
    assert x.fmt == 1 and x.c0 == 0 and x.c == x.ld
    out = Act.empty(x.n, x.h, x.w, x.c, x.buf.device, 0)
    N.check(N.lib().fcp_split32_to_f32(x.ptr(), out.ptr(), x.n * x.h * x.w, x.c, N.stream_ptr()), "fcp_split32_to_f32")
    return out
