"""``Cropper`` — same constructor, attributes and ``process_dir`` / ``process_batch`` /
``crop_align`` surface as the reference (cropper.py:139-156, :441-447, :748, :852-857),
with the batch kept resident on the GPU from the upload to the final crops:
detect -> [enhance] -> estimate + warp -> [parse] run as HIP kernels; the host sees
only the uint8 crops / masks it has to write to disk.
"""
from __future__ import annotations

import itertools
import os
from collections import Counter, defaultdict
from multiprocessing.pool import ThreadPool
from threading import BoundedSemaphore, Lock, local

import numpy as np
import torch

from . import align, trace
from .batch import build_batch
from .utils import get_ldm_slices, parse_landmarks_file, read_image, read_images, write_image


def landmarks_target(output_size, face_factor):
    """Target 5-point set (cropper.py:423-439): float32 table scaled in place."""
    std = align.STANDARD_LANDMARKS_5.copy()
    std[:, 0] *= output_size[0] * face_factor
    std[:, 1] *= output_size[1] * face_factor
    std[:, 0] += (1 - face_factor) * output_size[0] / 2
    std[:, 1] += (1 - face_factor) * output_size[1] / 2
    return std


class Cropper:
    def __init__(
        self,
        output_size: int | tuple[int, int] | list[int] = 256,
        output_format: str | None = None,
        resize_size: int | tuple[int, int] | list[int] = 1024,
        face_factor: float = 0.65,
        strategy: str = "largest",
        padding: str = "constant",
        allow_skew: bool = False,
        landmarks: str | tuple[np.ndarray, np.ndarray] | None = None,
        attr_groups: dict[str, list[int]] | None = None,
        mask_groups: dict[str, list[int]] | None = None,
        det_threshold: float | None = 0.6,
        enh_threshold: float | None = None,
        batch_size: int = 8,
        num_processes: int = 1,
        device: str | torch.device = "cuda:0",
        weights: dict | None = None,
        precision: str | None = None,
    ):
        """Arguments as in the reference (cropper.py:139-156).  ``device`` must be a GPU
        (``"cuda:N"``); ``weights`` optionally maps "retinaface"/"rrdb"/"bisenet" to a
        state dict / path / "generated" (default: the real checkpoints, from ``$FCP_WEIGHTS_DIR`` or the
        torch hub cache, else downloaded like the reference does; there is no silent random-weight fallback)."""
        self.output_size = output_size
        self.output_format = output_format
        self.resize_size = resize_size
        self.face_factor = face_factor
        self.strategy = strategy
        self.padding = padding
        self.allow_skew = allow_skew
        self.landmarks = landmarks
        self.attr_groups = attr_groups
        self.mask_groups = mask_groups
        self.det_threshold = det_threshold
        self.enh_threshold = enh_threshold
        self.batch_size = batch_size
        self.num_processes = num_processes
        self.device = device
        self.weights = weights or {}
        self.precision = precision   # "f16x3" (default) | "f32": arithmetic of the conv engine
        self.num_std_landmarks = 5
        # GPU worker threads of process_dir (each runs whole batches: upload, detect, align, read-back).  The reference's
        # `num_processes` is the size of its ThreadPool (cropper.py:900-902, default 1); here one worker leaves the device idle
        # while it is in its host phases — 2365-2438 images/s end to end against 3148-3209 with two (profiles/r06_probes.md
        # section 2) — so at least two batches (three for batch_size <= 8) are in flight unless told otherwise (gpu_workers = 1, or FCP_GPU_WORKERS=1).
        # The output set does not depend on it (tests/test_cropper_gpu.py::test_process_dir_pipeline_is_deterministic).
        self.gpu_workers = int(os.environ["FCP_GPU_WORKERS"]) if os.environ.get("FCP_GPU_WORKERS") else None
        # host I/O threads of process_dir (decode prefetch + asynchronous encode/write around the GPU workers)
        self.io_threads = max(2, min(16, (os.cpu_count() or 4) // 2))
        # ... and, by default, one decode / encode worker PROCESS behind every I/O thread (_io_pool.py): the
        # Pillow work leaves the parent's interpreter lock.  (readers, writers); None = sized from the host's cores;
        # FCP_IO_PROCESSES=0 (or io_processes = (0, 0)) keeps decode / encode on the threads
        self.io_processes = (0, 0) if os.environ.get("FCP_IO_PROCESSES", "1") == "0" else None
        self.io_ring_mb = None       # shared-memory ring of a decode worker in MiB (None: FCP_IO_RING_MB, default 128)
        self._io_procs = None
        # set by process_dir as ONE tuple (executor the encoded files are written on, futures of the writes still in
        # flight, semaphore bounding them), so that a task of a failed run can never see a half-reset state
        self._io = None
        self._write_lock = Lock()

        if isinstance(self.output_size, int):
            self.output_size = (self.output_size, self.output_size)
        if len(self.output_size) == 1:
            self.output_size = (self.output_size[0], self.output_size[0])
        if isinstance(self.resize_size, int):
            self.resize_size = (self.resize_size, self.resize_size)
        if len(self.resize_size) == 1:
            self.resize_size = (self.resize_size[0], self.resize_size[0])
        if isinstance(self.device, str):
            self.device = torch.device("cuda:0" if device in ("cuda", "hip") else device.replace("hip", "cuda"))
        if isinstance(self.landmarks, str):
            self.landmarks = parse_landmarks_file(self.landmarks)

        self._init_models()
        self._init_landmarks_target()

    # ------------------------------------------------------------------ init
    def _init_models(self):
        """cropper.py:346-390.  One immutable model set per Cropper (the reference
        re-creates them in every pool thread on the shared ``self``)."""
        self.det_model = None
        self.enh_model = None
        self.par_model = None
        if self.device.type != "cuda":
            raise RuntimeError("face_crop_plus_amd: device must be an AMD GPU ('cuda:N'); no CPU fallback")
        if self.device.index is not None:
            torch.cuda.set_device(self.device.index)
        if self.det_threshold is not None and self.landmarks is None:
            from .retinaface import RetinaFace
            self.det_model = RetinaFace(self.strategy, self.det_threshold)
            self.det_model.load(self.device, self.weights.get("retinaface"), self.precision)
        if self.enh_threshold is not None:
            from .rrdb import RRDBNet
            self.enh_model = RRDBNet(self.enh_threshold)
            self.enh_model.load(self.device, self.weights.get("rrdb"), self.precision)
        if self.attr_groups is not None or self.mask_groups is not None:
            from .bise import BiSeNet
            self.par_model = BiSeNet(self.attr_groups, self.mask_groups, self.batch_size)
            self.par_model.load(self.device, self.weights.get("bisenet"), self.precision)

    def _init_landmarks_target(self):
        if self.num_std_landmarks != 5:
            raise ValueError(f"Unsupported number of standard landmarks for estimating alignment transform "
                             f"matrix: {self.num_std_landmarks}.")
        self.landmarks_target = landmarks_target(self.output_size, self.face_factor)

    # ------------------------------------------------------------ crop_align
    def _crop_align_device(self, images_dev, paddings, indices, landmarks_dev):
        """Device tensors in, (crops (F,oh,ow,3) u8 device, ok (F,) i32 device) out."""
        pads = None if paddings is None else torch.as_tensor(np.asarray(paddings), dtype=torch.int32)
        idx = indices if isinstance(indices, torch.Tensor) else torch.as_tensor(np.asarray(indices), dtype=torch.int32)
        crops, ok, _ = align.crop_align(images_dev, idx, landmarks_dev, self.landmarks_target, self.output_size,
                                        align.border_code(self.padding), self.allow_skew, pads)
        return crops, ok

    def crop_align(self, images, padding, indices, landmarks_source) -> np.ndarray:
        """Reference signature (cropper.py:441-552): numpy in, numpy out.  ``images`` is an
        (N,H,W,3) uint8 array or a list of differently-sized arrays."""
        if len(indices) == 0:
            return np.array([])
        lms = torch.from_numpy(np.ascontiguousarray(landmarks_source, dtype=np.float32))
        outs = []
        if isinstance(images, np.ndarray) and images.ndim == 4:
            dev_imgs = torch.from_numpy(np.ascontiguousarray(images)).to(self.device)
            crops, ok = self._crop_align_device(dev_imgs, padding, list(indices), lms.to(self.device))
            crops, ok = crops.cpu().numpy(), ok.cpu().numpy()
            outs = [c for c, o in zip(crops, ok) if o]
        else:
            # ragged list: one upload + one launch per source image, face order preserved
            order = defaultdict(list)
            for li, ii in enumerate(indices):
                order[int(ii)].append(li)
            res = {}
            for ii, lis in order.items():
                dev_img = torch.from_numpy(np.ascontiguousarray(images[ii])).to(self.device)[None]
                pad = None if padding is None else np.asarray(padding)[ii:ii + 1]
                crops, ok = self._crop_align_device(dev_img, pad, [0] * len(lis), lms[lis].to(self.device))
                for li, c, o in zip(lis, crops.cpu().numpy(), ok.cpu().numpy()):
                    if o:
                        res[li] = c
            outs = [res[li] for li in sorted(res)]
        return np.stack(outs) if len(outs) > 0 else np.array(outs)

    # ----------------------------------------------------------------- saving
    MAX_PENDING_WRITES = 256     # encode / write tasks in flight before a GPU worker waits (process_dir)

    def _target_paths(self, file_names, output_dir: str):
        """Where each face of one group goes, in face order.  Naming rules of the reference's writer
        (cropper.py:588-603): the source file's stem; with strategy "all" a running "_<k>" per source file,
        counted inside this group, starting at 0; the source extension unless ``output_format`` overrides it."""
        nth = Counter()
        paths = []
        for source in map(str, file_names):
            stem, ext = os.path.splitext(source)
            if self.output_format is not None:
                ext = "." + self.output_format
            if self.strategy == "all":
                stem = f"{stem}_{nth[source]}"
                nth[source] += 1
            paths.append(os.path.join(output_dir, stem + ext))
        return paths

    def _emit(self, path: str, pixels: np.ndarray):
        """Write one file: inline, or — inside process_dir — as a task on the I/O pool.  At most
        MAX_PENDING_WRITES tasks are in flight: a slow disk stalls the GPU worker here instead of piling uint8
        crops up in host memory, and a failed write surfaces at the next batch, not at the end of the run."""
        # Locals: process_dir resets the attributes when it unwinds, while tasks of a failed run may still be in flight.
        writer, writes, slots = getattr(self, "_io", None) or (None, None, None)      # ONE read: never a torn triple
        if writer is None:
            write_image(path, pixels)
            return
        slots.acquire()
        procs = getattr(self, "_io_procs_active", None)       # encode in the thread's worker process, or on the thread

        def task():
            try:
                if procs is not None:
                    procs.write(path, pixels)
                else:
                    write_image(path, pixels)
            finally:
                slots.release()
        with self._write_lock:
            done = [w for w in writes if w.done()]
            writes[:] = [w for w in writes if not w.done()]
            try:
                writes.append(writer.submit(task))
            except BaseException:
                slots.release()          # the task will never run: give its slot back
                raise
        for w in done:
            w.result()               # re-raise an encode / write error of an earlier file now

    def save_group(self, faces, file_names, output_dir: str):
        """One directory of faces (or masks) — reference ``save_group``, cropper.py:554-609, with Pillow as the
        encoder (arrays are RGB already, so there is no colour swap)."""
        if len(faces) == 0:
            return
        os.makedirs(output_dir, exist_ok=True)
        for path, pixels in zip(self._target_paths(file_names, output_dir), faces):
            self._emit(path, np.asarray(pixels))

    def save_groups(self, faces, file_names, output_dir, attr_groups, mask_groups):
        """Directory tree ``output_dir/<attr group>/<mask group>[_mask]`` — reference ``save_groups``,
        cropper.py:611-746.  A face lands in every (attr, mask) cell both groups list it in; the masks of a
        cell go to the sibling ``<mask group>_mask`` directory under the same file names."""
        everyone = list(range(len(faces)))
        attrs = {"": everyone} if attr_groups is None else attr_groups
        masks = {"": (everyone, None)} if mask_groups is None else mask_groups
        for (attr_name, in_attr), (mask_name, (in_mask, mask_rows)) in itertools.product(attrs.items(), masks.items()):
            # Order matters (it decides which face of a file gets which "_<k>"): the reference iterates the
            # CPython set `set(a) & set(b)`, so the very same expression is evaluated here.
            cell = list(set(in_attr) & set(in_mask))
            cell_dir = os.path.join(output_dir, attr_name, mask_name)
            sources = file_names[cell]
            self.save_group([faces[i] for i in cell], sources, cell_dir)
            if mask_rows is not None:
                row_of = {}
                for row, face in enumerate(in_mask):
                    row_of.setdefault(face, row)
                self.save_group(mask_rows[[row_of[i] for i in cell]], sources, cell_dir + "_mask")

    # ------------------------------------------------------------- processing
    def _landmark_rows(self, table_names):
        """file name -> rows of the user's landmark table, built once per table (not per batch)."""
        cached = getattr(self, "_rows_cache", None)
        if cached is None or cached[0] is not table_names:
            rows_of = defaultdict(list)
            for row, name in enumerate(table_names):
                rows_of[str(name)].append(row)
            cached = self._rows_cache = (table_names, rows_of)
        return cached[1]

    @torch.no_grad()
    def process_batch(self, file_names, input_dir: str, output_dir: str):
        """cropper.py:748-850."""
        images, file_names = read_images(file_names, input_dir)
        self._process_images(images, file_names, output_dir)

    @torch.no_grad()
    def _process_images(self, images, file_names, output_dir: str, pinned=None):
        """Everything of ``process_batch`` after the files have been decoded.  ``pinned``: per-image flags of arrays that
        live in page-locked memory (``build_batch`` uploads those without a staging copy)."""
        if len(images) == 0:
            return
        paddings, landmarks, indices, images_dev = None, None, None, None
        with torch.cuda.device(self.device):
            if self.landmarks is None and self.det_model is None:
                indices = list(range(len(file_names)))              # one "face" per image, no alignment
            elif self.landmarks is not None:
                # user-supplied landmark sets (cropper.py:796-813): rows of the landmark table whose file name is in
                # this batch, grouped by image in batch order; images without a row are dropped
                table, table_names = self.landmarks
                rows_of = self._landmark_rows(table_names)
                pairs = [(i, row) for i, name in enumerate(file_names) for row in rows_of.get(str(name), ())]
                indices = [i for i, _ in pairs]
                landmarks = table[[row for _, row in pairs]]
            else:
                with trace.range("fcp:build_batch"):
                    images_dev, _, paddings = build_batch(images, self.resize_size, "constant", self.device, pinned)
                with trace.range("fcp:detect"):
                    lm_np, indices = self.det_model.predict(images_dev)
                landmarks = lm_np - paddings[indices][:, None, [2, 0]].astype(np.float32) if len(indices) else lm_np

            if landmarks is not None and len(landmarks) == 0:
                return
            if landmarks is not None and landmarks.shape[1] != self.num_std_landmarks:
                slices = get_ldm_slices(self.num_std_landmarks, landmarks.shape[1])
                landmarks = np.stack([landmarks[:, s].mean(1) for s in slices], 1)

            if self.enh_model is not None:
                with trace.range("fcp:enhance"):
                    if images_dev is not None:
                        images_dev = self.enh_model.predict(images_dev, landmarks, indices)
                    else:
                        # ragged list (no detector): the reference hands the whole list to RRDBNet.predict
                        # (cropper.py:833-836), whose gate measures every image's faces against the area of
                        # images[0] (rrdb.py:124-140) and skips images that have no landmark set
                        todo = self.enh_model.gate(len(images), images[0].shape[0], images[0].shape[1], landmarks, indices)
                        for i in todo:
                            images[i] = self.enh_model.predict(torch.from_numpy(images[i]).to(self.device)[None],
                                                               None, None)[0].cpu().numpy()

            groups = (None, None)
            if landmarks is not None:
                if images_dev is not None:
                    with trace.range("fcp:align"):
                        crops_dev, ok = self._crop_align_device(
                            images_dev, paddings, list(indices),
                            torch.from_numpy(np.ascontiguousarray(landmarks, dtype=np.float32)).to(self.device))
                    keep = ok.cpu().numpy() != 0
                    crops_dev = crops_dev[torch.from_numpy(keep).to(self.device)]
                    indices = [i for i, k in zip(indices, keep) if k]
                    faces_dev, faces = crops_dev, crops_dev.cpu().numpy()
                else:
                    with trace.range("fcp:align"):
                        faces = self.crop_align(images, paddings, indices, landmarks)
                    faces_dev = torch.from_numpy(faces).to(self.device) if len(faces) else None
            else:
                # no alignment: the decoded images themselves are the "faces".  Inside process_dir they may be views of a
                # decode worker's shared-memory ring, which is recycled as soon as this call returns, while the encode
                # tasks run later: they get their own copies
                faces, faces_dev = ([np.array(im) for im in images] if pinned is not None else images), None
            if self.par_model is not None and len(faces) > 0:
                if faces_dev is None:
                    faces_dev = [torch.from_numpy(np.ascontiguousarray(f)).to(self.device) for f in faces]
                with trace.range("fcp:parse"):
                    groups = self.par_model.predict(faces_dev)
        with trace.range("fcp:save"):
            self.save_groups(faces, file_names[indices], output_dir, *groups)

    def process_dir(self, input_dir: str, output_dir: str | None = None, desc: str | None = "Processing"):
        """cropper.py:852-909: batches of file names over a thread pool sharing the models."""
        if output_dir is None:
            output_dir = input_dir + "_faces"
        files, bs = os.listdir(input_dir), self.batch_size
        files = sorted(files)          # deterministic batches, so ranks agree on the partition
        file_batches = [files[i:i + bs] for i in range(0, len(files), bs)]
        from .dist import shard
        file_batches = shard(file_batches)   # rank r of R takes batches r, r+R, ... (no-op single-process)
        if len(file_batches) == 0:
            return
        # Overlapped host I/O: decode is prefetched `depth` batches ahead on an I/O pool, the
        # `num_processes` GPU workers of the reference's ThreadPool only run the device pipeline, and the
        # encoded crops / masks are written asynchronously on the same I/O pool.  File naming, warn-and-skip
        # and the output directory layout are exactly those of the synchronous `process_batch`.
        from concurrent.futures import ThreadPoolExecutor
        # small batches (the reference's default batch_size = 8) leave the device idle even with two in flight: 1264-1282 images/s
        # end to end with three workers against 1086-1258 with two (batch 8 @1024^2), while batch 32 prefers two (1357-1368 against 1261-1289)
        workers = max(1, self.gpu_workers) if self.gpu_workers else max(3 if self.batch_size <= 8 else 2, self.num_processes)
        depth = max(2, 2 * workers)
        procs = self._io_processes()
        if procs is not None:
            # one I/O thread per worker process (a thread only relays: request, blocking reply); decode and encode have
            # their own executors so that a burst of reads can never starve the writes a batch needs to finish
            procs.begin()
            io = ThreadPoolExecutor(max_workers=procs.readers, thread_name_prefix="fcp-read")
            wio = ThreadPoolExecutor(max_workers=procs.writers, thread_name_prefix="fcp-write")
            read_one = None
        else:
            io = wio = ThreadPoolExecutor(max_workers=self.io_threads, thread_name_prefix="fcp-io")
            read_one = lambda path: (read_image(path), None)
        self._io_procs_active = procs
        writes = []
        self._io = (wio, writes, BoundedSemaphore(self.MAX_PENDING_WRITES))
        # Threads: every file is its own decode task (a batch decoded by one thread would cap the pipeline at `depth`
        # decoders).  Worker processes: a batch is dealt round-robin into one request per decoder — the parent's relay
        # threads wake up once per request, not once per image (their wake-ups contend for the interpreter lock).
        def submit_read(i):
            paths = [os.path.join(input_dir, f) for f in file_batches[i]]
            if procs is None:
                return [io.submit(read_one, p_) for p_ in paths]
            k = min(procs.readers, len(paths))
            return [(paths[j::k], io.submit(procs.read_many, paths[j::k])) for j in range(k)]

        def collect_read(i, futs):
            """-> images, surviving names, release tokens of the shared-memory regions the images live in."""
            if procs is None:
                decoded = [f.result() for f in futs]
            else:                                    # undo the round-robin deal: image m of the batch is item m // k of chunk m % k
                chunks = [f.result() for _, f in futs]
                decoded = [chunks[m % len(chunks)][m // len(chunks)] for m in range(len(file_batches[i]))]
            ok = [k for k, (im, _) in enumerate(decoded) if im is not None]
            return [decoded[k][0] for k in ok], np.array(file_batches[i])[ok], [decoded[k][1] for k in ok]

        reads = {i: submit_read(i) for i in range(min(depth, len(file_batches)))}
        lock = Lock()

        tls = local()

        def worker(i):
            futs = reads.pop(i, None)
            with lock:
                nxt = i + depth
                if nxt < len(file_batches) and nxt not in reads:
                    reads[nxt] = submit_read(nxt)
            images, names, tokens = collect_read(i, futs) if futs is not None else (*read_images(file_batches[i], input_dir), [])
            pinned = procs.pinned_flags(tokens) if procs is not None and tokens else None   # images in page-locked rings
            try:
                if workers == 1:
                    return self._process_images(images, names, output_dir, pinned)
                # one HIP stream per GPU worker: the batches of different workers overlap on the device (the tail of
                # one kernel with the head of another; measured +4 % at two streams) instead of queueing on stream 0
                if not hasattr(tls, "stream"):
                    from . import engine as E
                    with torch.cuda.device(self.device):
                        tls.stream = E.thread_main_stream(self.device)           # the same streams again in every run
                        tls.stream.wait_stream(torch.cuda.default_stream())      # filters were uploaded there
                with torch.cuda.device(self.device), torch.cuda.stream(tls.stream):
                    self._process_images(images, names, output_dir, pinned)
                    tls.stream.synchronize()
            finally:
                # every use of the decoded images is over — the uploads out of the rings were enqueued before kernels whose
                # results _process_images has read back (a stream synchronisation), the no-alignment path copied what it
                # hands to the asynchronous writers — so their ring regions go back to the decode workers
                del images
                if procs is not None:
                    procs.release(tokens)

        failed = True
        try:
            with ThreadPool(workers) as pool:
                imap = pool.imap(worker, range(len(file_batches)))
                if desc is not None:
                    try:
                        import tqdm
                        imap = tqdm.tqdm(imap, total=len(file_batches), desc=desc)
                    except ImportError:
                        pass
                list(imap)
            for w in writes:
                w.result()                       # surface encode / write errors
            failed = False
        finally:
            io.shutdown(wait=True)       # in-flight tasks still hold the semaphore / list: reset only afterwards
            if wio is not io:
                wio.shutdown(wait=True)
            self._io = None
            self._io_procs_active = None
            # A run that failed may have left a dead worker, a request cut off in the middle (its ring space handed out but
            # never reported) or prefetched batches nobody collected: that pool is not reused — the next run starts fresh
            # workers instead of inheriting the damage.
            if procs is not None and (failed or not procs.healthy()):
                procs.close()
                if self._io_procs is procs:
                    self._io_procs = None

    @staticmethod
    def _pin_ring(ring, nbytes) -> bool:
        """Page-lock (nbytes > 0) / unlock (0) a decode ring for HIP, so that ``build_batch`` uploads images straight from
        the ring instead of through a host copy into its staging blob (14 of 40 ms per batch of 64 at 640^2)."""
        addr = np.frombuffer(ring, dtype=np.uint8).ctypes.data
        rt = torch.cuda.cudart()
        try:
            if nbytes:
                err = rt.cudaHostRegister(addr, nbytes, 0)
                ok = int(err) == 0
                return ok and torch.from_numpy(np.frombuffer(ring, dtype=np.uint8, count=64)).is_pinned()
            rt.cudaHostUnregister(addr)
        except Exception:                            # noqa: BLE001 - a runtime without host registration: staging still works
            pass
        return False

    @staticmethod
    def default_io_processes(cores: int, local_world: int = 1):
        """(decode, encode) worker processes of one ``Cropper`` on a host with ``cores`` usable cores shared by
        ``local_world`` ranks (one process per GPU: the ranks of a node share the host).  Measured on a 2 x 64-core EPYC
        9575F box (2048 JPEGs of 640^2, one decode request per decoder and batch; 2 / 3 GPU workers): (8, 3) 2320 / 2650,
        (12, 3) 2360 / 2750, (12, 4) 2300 / 2540, (16, 4) 2270 / 2600, (24, 4) 2110 / 2650 images/s — flat beyond a dozen
        decoders (one decodes ~600 images/s).  8 ranks on that box: 16 cores each -> (5, 2), i.e. 40 + 16 I/O processes,
        8 x (1 + num_processes) Python threads that enqueue, and 3000 decodes/s per rank against ~3500 faces/s of device
        rate."""
        cores = max(1, cores // max(1, local_world))
        return (max(2, min(12, cores // 3)), max(1, min(3, cores // 8)))

    def _io_processes(self):
        """The decode / encode worker processes of this Cropper (started on first use, reused by later runs), or None
        when they are switched off or cannot be had on this platform."""
        want = self.io_processes
        if want is None:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
            want = self.default_io_processes(cores, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        if min(want) <= 0:
            return None
        have = self._io_procs
        if have is not None and have.healthy() and (have.readers, have.writers) == tuple(want):
            return have
        if have is not None:
            have.close()
        try:
            from ._io_pool import IOProcesses
            kw = {} if getattr(self, "io_ring_mb", None) is None else {"ring_mb": int(self.io_ring_mb)}
            self._io_procs = IOProcesses(*want, register=self._pin_ring if os.environ.get("FCP_IO_PIN", "1") != "0" else None, **kw)
        except (ValueError, OSError) as e:           # no memfd / no processes left on this host: threads still work
            import warnings
            warnings.warn(f"decode / encode worker processes unavailable ({e}): using I/O threads")
            self.io_processes, self._io_procs = (0, 0), None
        return self._io_procs
