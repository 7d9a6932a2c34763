"""Worker PROCESSES that decode image files straight into shared, page-locked staging memory -- the host side of
``TensorFlowInference.extract_files`` (the loop of facerec_test.py:394 at gallery scale).

Why processes: ``Image.open`` / ``convert`` / ``np.asarray`` hold the GIL for most of a 250x250 JPEG, so a thread pool of
PIL decoders tops out at ~1.5x one core (3.1 k img/s on 32 threads where one thread does 0.84 k).
Each worker here is a plain Python process (spawned, never forked from a process that holds a HIP context; it imports
PIL + NumPy only, no torch, no GPU) that receives (paths, region) tasks, decodes exactly as ``preprocess.imread_rgb``
does (``misc.imread(path, mode='RGB')``, facerec_test.py:83,91) and packs the pixels back to back into its region of a
``multiprocessing.shared_memory`` block.  The parent registers that block with the HIP runtime (``hipHostRegister``), so
the upload is an asynchronous DMA out of the very bytes the decoder wrote: no pickling of pixels, no copy into a second
pinned buffer, no per-(H, W) staging buffers (staging is a flat byte pool per slot, sized by capacity).

The pool is persistent (workers start once per extractor, ~0.3 s) and is closed by ``close_session()``.
"""
from __future__ import annotations

import os
import sys
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np


_MAIN_PATCH_LOCK = threading.Lock()       # DecodePool.__init__ edits sys.modules['__main__'] for a moment: one thread at a time


def cpu_quota() -> Optional[float]:
    """CPUs' worth of time the container may use per period (cgroup v2 ``cpu.max``, v1 ``cpu.cfs_quota_us``), or None if unlimited.
    The affinity mask does not show it: the GPU boxes of round 5 list 256 CPUs and run under ``1600000 100000`` = 16 CPUs -- a pool of
    32 decoders there decodes exactly what 16 do (34-36 k photos/s) and is throttled in half of the scheduler's periods."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            q = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            per = float(fh.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def default_workers() -> int:
    """Decoder processes for ONE extractor: the cores this process may use -- the affinity mask capped by the container's CPU
    QUOTA (cpu_quota) -- SHARED between the ranks of the node (``LOCAL_WORLD_SIZE``, set by torchrun: eight ranks of a file-driven
    gallery job on one host get an eighth of the cores each, not 32 decoders apiece), at most 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q is not None:
        n = max(1, min(n, int(q)))
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    return max(1, min(n // local_world, 32))


def cpu_order() -> List[int]:
    """The CPUs this process may use, ordered so that DISTINCT PHYSICAL CORES come first (one hardware thread of every core, then the
    second threads): worker k of a pool is pinned to entry k of this list, so up to one worker per core never shares a core with another
    decoder, and a worker never migrates (its decoder's tables stay in its core's caches).  Falls back to the affinity mask in numeric
    order where the topology files are not readable."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return list(range(os.cpu_count() or 1))
    firsts, rest, seen = [], [], set()
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as fh:
                sib = fh.read().strip()
        except OSError:
            sib = str(c)
        if sib in seen:
            rest.append(c)
        else:
            seen.add(sib)
            firsts.append(c)
    return firsts + rest


_RUNS_LOCK = threading.Lock()
_LIVE_RUNS = set()                        # runs of cpu_order() held by the pinned pools alive in this process


def pin_offset(workers: int) -> Optional[int]:
    """First entry of cpu_order() for this process's pinned decoders, from what a launcher says about our place on the node:
    HSEFR_DECODE_CPU_OFFSET (explicit) plus LOCAL_RANK (torchrun) or SLURM_LOCALID (srun) times the pool size.  None when the
    environment says nothing -- the caller then does not pin by default."""
    off, known = 0, False
    for key, scale in (("HSEFR_DECODE_CPU_OFFSET", 1), ("LOCAL_RANK", workers), ("SLURM_LOCALID", workers)):
        v = os.environ.get(key)
        if v is None or (key == "SLURM_LOCALID" and "LOCAL_RANK" in os.environ):
            continue
        try:
            off += max(0, int(v)) * scale
            known = True
        except ValueError:
            pass
    return off if known else None


def _worker_main(shm_name: str, tasks, results, cpu: Optional[int] = None) -> None:
    """Runs in the worker process.  task = (task_id, [paths], byte offset, byte capacity) or None to stop.
    result = (task_id, [(offset, H, W) | None per image], spill, error): images that did not fit the region are returned
    pickled in ``spill`` {position: ndarray}; ``error`` = (exception type name, message) of the first failing file.
    cpu: the CPU this worker pins itself to (None: wherever the scheduler puts it)."""
    import signal
    from multiprocessing import shared_memory
    signal.signal(signal.SIGINT, signal.SIG_IGN)          # the parent decides when the pool stops
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except (AttributeError, OSError):
            pass
    from PIL import Image
    shm = shared_memory.SharedMemory(name=shm_name)
    buf = np.frombuffer(shm.buf, dtype=np.uint8)
    try:
        while True:
            task = tasks.get()
            if task is None:
                break
            tid, paths, off, cap = task
            metas, spill, err = [], {}, None
            end = off + cap
            for k, p in enumerate(paths):
                try:
                    with Image.open(p) as im:             # == preprocess.imread_rgb
                        a = np.asarray(im.convert("RGB"))
                except Exception as e:                    # reported to the parent, which raises it (first failing file)
                    try:                                  # the exception OBJECT when it pickles (PIL.UnidentifiedImageError,
                        import pickle                     # OSError with errno ...): the parent raises what the serial path raises
                        blob = pickle.dumps(e)
                        pickle.loads(blob)
                    except Exception:
                        blob = None
                    err = (type(e).__name__, "%s" % (e,), p, blob)
                    break
                nb = a.size
                if off + nb <= end:
                    buf[off:off + nb] = a.reshape(-1)
                    metas.append((off, a.shape[0], a.shape[1]))
                    off += nb
                else:
                    metas.append(None)
                    spill[k] = a
            results.put((tid, metas, spill, err))
    finally:
        del buf
        shm.close()


def _rebuild_error(name: str, msg: str, path: str, blob: Optional[bytes]) -> BaseException:
    """The worker's exception in the parent: the pickled object itself when it travelled (same type -- also non-builtin ones
    such as PIL.UnidentifiedImageError, an OSError the serial imread_rgb raises -- same args, errno, filename); otherwise
    a builtin of that name built from the message, and RuntimeError when even that constructor refuses one argument."""
    if blob is not None:
        try:
            import pickle
            e = pickle.loads(blob)
            if isinstance(e, Exception):
                return e
        except Exception:
            pass
    import builtins
    exc = getattr(builtins, name, None)
    if exc is FileNotFoundError:
        return FileNotFoundError(2, "No such file or directory", path)
    if isinstance(exc, type) and issubclass(exc, Exception):
        try:
            return exc(msg)
        except TypeError:               # e.g. UnicodeDecodeError: five positional arguments
            pass
    return RuntimeError("%s: %s (%s)" % (name, msg, path))


class DecodePool:
    """``slots`` staging slots of ``slot_bytes`` each in one shared block; a chunk of files is decoded into one slot by
    tasks of ``task_files`` files, each task owning a fixed sub-region (so workers never contend for space)."""

    def __init__(self, workers: Optional[int] = None, slot_bytes: int = 64 << 20, slots: int = 3, task_files: int = 8,
                 pin: Optional[bool] = None):
        """pin: every worker pins itself to one CPU of cpu_order() -- distinct physical cores first -- starting at the offset
        pin_offset() derives from the launcher's environment: under torchrun / srun the ranks of a node take consecutive runs of that
        list, so eight ranks' decoders do not sit on the same cores.  None (default): pin only when such an offset is KNOWN
        (LOCAL_RANK, SLURM_LOCALID or HSEFR_DECODE_CPU_OFFSET is set) -- processes started by hand (eight jobs with
        HIP_VISIBLE_DEVICES, say) would otherwise all pin to the same first cores, and the scheduler spreads them better than that
        (ADVICE r5).  True: pin regardless (a single job that owns the host: bench.py).  Pools alive in ONE process take consecutive
        runs of the list either way."""
        import multiprocessing as mp
        from multiprocessing import shared_memory
        self.workers = int(workers or default_workers())
        first = pin_offset(self.workers)
        if pin is None:
            pin = first is not None
        cpus = cpu_order() if pin else []
        self._run = None
        if cpus:
            with _RUNS_LOCK:                                   # the lowest run of `workers` CPUs no live pool of this process holds
                self._run = next(r for r in range(len(_LIVE_RUNS) + 1) if r not in _LIVE_RUNS)
                _LIVE_RUNS.add(self._run)
            first = (first or 0) + self._run * self.workers
        self.cpus = [cpus[(first + k) % len(cpus)] for k in range(self.workers)] if cpus else [None] * self.workers
        self.slot_bytes, self.slots, self.task_files = int(slot_bytes), int(slots), int(task_files)
        self._ctx = mp.get_context("spawn")
        self._shm = shared_memory.SharedMemory(create=True, size=self.slot_bytes * self.slots)
        self._np = np.frombuffer(self._shm.buf, dtype=np.uint8)
        self._tasks = self._ctx.Queue()
        self._results = self._ctx.Queue()
        self._procs = []
        # spawn re-imports __main__ in the child when it has a file (multiprocessing's "main path" fix-up): keep a heavy main
        # module (bench.py, pytest) out of the decoders -- they need this module only
        with _MAIN_PATCH_LOCK:
            main = sys.modules.get("__main__")
            saved = getattr(main, "__file__", None), getattr(main, "__spec__", None)
            try:
                if main is not None:
                    if saved[0] is not None:
                        del main.__file__
                    main.__spec__ = None
                for k in range(self.workers):
                    p = self._ctx.Process(target=_worker_main, args=(self._shm.name, self._tasks, self._results, self.cpus[k]), daemon=True)
                    p.start()
                    self._procs.append(p)
            finally:
                if main is not None:
                    if saved[0] is not None:
                        main.__file__ = saved[0]
                    main.__spec__ = saved[1]
        self._pinned = False
        self._tensor = None
        self._next_tid = 0
        import atexit
        import weakref
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())     # an extractor nobody closed: stop the workers, unlink the block
        self._open = {}            # chunk key -> {"tasks": {tid: (first file position, n files)}, "done": {tid: result}}
        self._stash = {}           # results that arrived for another chunk

    # -- staging memory as a torch tensor (page-locked when the runtime lets us) -------------------------------------
    def tensor(self):
        """uint8 CPU tensor over the whole shared block; registered with the HIP runtime once (hipHostRegister) so
        ``.to(device, non_blocking=True)`` from it is a true asynchronous copy."""
        if self._tensor is None:
            import torch
            self._tensor = torch.frombuffer(self._shm.buf, dtype=torch.uint8)
            try:
                rc = torch.cuda.cudart().cudaHostRegister(self._tensor.data_ptr(), self._tensor.numel(), 0)
                self._pinned = (int(rc) == 0)
            except Exception:
                self._pinned = False
        return self._tensor

    @property
    def pinned(self) -> bool:
        return self._pinned

    # -- chunks ----------------------------------------------------------------------------------------------------
    def submit(self, key, paths: Sequence[str], slot: int) -> None:
        """Queue the decode of ``paths`` into staging slot ``slot``; collect with ``collect(key)``."""
        n = len(paths)
        ntask = max(1, -(-n // self.task_files))
        region = (self.slot_bytes // ntask) & ~63
        base = slot * self.slot_bytes
        tasks = {}
        for t in range(ntask):
            lo, hi = t * self.task_files, min((t + 1) * self.task_files, n)
            tid = self._next_tid
            self._next_tid += 1
            tasks[tid] = (lo, hi - lo)
            self._tasks.put((tid, list(paths[lo:hi]), base + t * region, region))
        self._open[key] = {"tasks": tasks, "n": n}

    def collect(self, key) -> List[Tuple[int, Optional[Tuple[int, int, int]], Optional[np.ndarray]]]:
        """Blocks until every task of the chunk is done.  Returns one (position, (offset, H, W) | None, spilled array | None)
        per file, in file order; raises the first decoding error as the exception type the serial path would raise."""
        import queue
        ch = self._open.pop(key)
        want = dict(ch["tasks"])
        done = {}
        for tid in list(want):
            if tid in self._stash:
                done[tid] = self._stash.pop(tid)
        while len(done) < len(want):
            try:
                res = self._results.get(timeout=5.0)
            except queue.Empty:
                dead = [p for p in self._procs if not p.is_alive()]
                if dead:
                    raise RuntimeError("a decoder process died (exit code %r)" % dead[0].exitcode)
                continue
            if res[0] in want:
                done[res[0]] = res
            else:
                self._stash[res[0]] = res
        out = []
        first_err = None
        for tid in sorted(want):
            lo, cnt = want[tid]
            _, metas, spill, err = done[tid]
            if err is not None and first_err is None:
                first_err = err
            for k, m in enumerate(metas):
                out.append((lo + k, m, spill.get(k)))
        if first_err is not None:
            raise _rebuild_error(*first_err)
        return out

    def close(self) -> None:
        if self._shm is None:
            return
        for _ in self._procs:
            try:
                self._tasks.put(None)
            except Exception:
                pass
        for p in self._procs:
            p.join(timeout=5.0)
            if p.is_alive():
                p.terminate()
        self._procs = []
        if self._run is not None:
            with _RUNS_LOCK:
                _LIVE_RUNS.discard(self._run)
            self._run = None
        if self._tensor is not None and self._pinned:
            try:
                import torch
                torch.cuda.cudart().cudaHostUnregister(self._tensor.data_ptr())
            except Exception:
                pass
        self._tensor = None
        self._np = None
        for q in (self._tasks, self._results):
            try:
                q.close()
                q.join_thread()
            except Exception:
                pass
        try:
            self._shm.close()
        except BufferError:       # a view is still alive somewhere: the block is unlinked anyway and dies with the process
            pass
        try:
            self._shm.unlink()
        except FileNotFoundError:
            pass
        self._shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
