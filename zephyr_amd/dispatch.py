"""In-process dispatch of independent work items (frequencies, source batches) over the GPUs of a node.

The reference's `BaseMPDist.__mul__` hands one sub-problem per frequency to a `multiprocessing.Pool`
(zephyr/backend/distributors.py:80-96,161-168) and yields the results in submission order.  The GPU
counterpart keeps everything in one process:

  * one SOLVE thread per device works through that device's items in order (ctypes releases the GIL
    inside libhelm, so the threads of different devices run concurrently);
  * one PREPARE thread per device stays `lookahead` items ahead of it: it builds the next operator
    (model upload, assembly on the GPU) and starts its factorisation with `helm_prefactor`, which only
    enqueues launches on a high-priority stream -- the latency-bound top of the elimination tree of
    frequency k+1 then runs underneath the bandwidth-bound triangular solves of frequency k;
  * results come back as futures, consumed in submission order by the caller's generator.

Nothing here touches device memory itself; an item is two callables supplied by the caller.
"""
import os
import queue
import threading
from concurrent.futures import Future


def visible_devices():
    """Devices an in-process dispatcher may use.  `HELM_DEVICES=0,2,3` restricts the list; under a
    one-process-per-GPU launcher (torch.distributed initialised, or LOCAL_RANK set) a process keeps to its own GPU."""
    from . import _lib
    from .discretization import default_device
    env = os.environ.get('HELM_DEVICES')
    if env:
        return [int(t) for t in env.replace(' ', '').split(',') if t != '']
    from . import parallel
    if parallel.rank_and_size()[1] > 1 or 'LOCAL_RANK' in os.environ or 'HELM_DEVICE' in os.environ:
        return [default_device()]
    n = _lib.load().helm_device_count()
    return list(range(max(1, n)))


def workers_per_device(default=1):
    """Pipelines per GPU (HELM_WORKERS_PER_DEVICE, else `default`).  More than one lets the device-to-host copy of one frequency's
    wavefields run while another worker's frequency is being solved: worth it for callers that take host arrays back (`MultiFreq * q`:
    with three there is always a finished frequency to copy, and the results arrive at the PCIe rate), pointless for device-resident
    ones."""
    try:
        return max(1, int(os.environ.get('HELM_WORKERS_PER_DEVICE', default)))
    except ValueError:
        return default


def results_ahead(default=2):
    'finished results a worker may hold before its consumer has taken any (HELM_RESULTS_AHEAD): the depth of its Throttle'
    try:
        return max(1, int(os.environ.get('HELM_RESULTS_AHEAD', default)))
    except ValueError:
        return default


class Throttle(object):
    """Back-pressure between a worker and the consumer of its results: at most `depth` results of a worker exist that the caller has
    not taken yet (a frequency's wavefields are GBs of pinned host memory; the reference's pool keeps every pending result alive).
    `close()` lifts the limit -- an abandoned generator must not leave worker threads waiting."""

    def __init__(self, depth=2):
        self._sem = threading.Semaphore(max(1, int(depth)))
        self._closed = False

    def acquire(self):
        while not self._closed:
            if self._sem.acquire(timeout=0.05):
                return

    def release(self):
        self._sem.release()

    def close(self):
        self._closed = True


class _QuietBlas(object):
    """While a pipeline runs, BLAS / OpenMP pools of the process are held to one thread (threadpoolctl, when it is installed): a threaded BLAS call made from
    a work item -- or by the caller while it consumes results -- leaves its workers spinning for ~100 ms, and under a container CPU quota that freezes every thread
    of the process, the ones feeding the GPU included, for most of a scheduler period (zephyr_amd/problem.py, _norm2).  Nested use counts; the limits the caller
    had come back when the last pipeline has joined.  Nothing here can stop workers that were already spinning when the pipeline started: set
    OPENBLAS_NUM_THREADS / OMP_NUM_THREADS for the process where a quota is in force."""
    _lock = threading.Lock()
    _depth = 0
    _limiter = None
    _controller = None

    @classmethod
    def enter(cls):
        with cls._lock:
            cls._depth += 1
            if cls._depth == 1 and os.environ.get('HELM_QUIET_BLAS', '1') != '0':
                try:
                    if cls._controller is None:        # (looking the process's BLAS / OpenMP libraries up costs 0.5 ms with torch loaded; the limit itself 0.03)
                        from threadpoolctl import ThreadpoolController
                        cls._controller = ThreadpoolController()
                    cls._limiter = cls._controller.limit(limits=1)
                except Exception:
                    cls._limiter = None

    @classmethod
    def leave(cls):
        with cls._lock:
            cls._depth = max(0, cls._depth - 1)
            if cls._depth == 0 and cls._limiter is not None:
                try:
                    cls._limiter.restore_original_limits()
                except Exception:
                    pass
                cls._limiter = None


class WorkItem(object):
    """prepare() runs on the device's prepare thread (may be None), solve(prepared) on its solve thread;
    `future` receives solve's return value or the first exception of either step."""

    __slots__ = ('prepare', 'solve', 'future', '_prepared', '_error', 'owner', 'nrow', 'ncol')

    def __init__(self, solve, prepare=None):
        self.prepare = prepare
        self.solve = solve
        self.future = Future()
        self._prepared = None
        self._error = None


class DevicePipeline(object):
    """Two threads for one GPU: items are prepared up to `lookahead` ahead of the one being solved."""

    def __init__(self, device, lookahead=1, strict=False, solvers=1, group=1, group_prepare=None):
        self.device = device
        self.lookahead = max(0, int(lookahead))
        self.strict = bool(strict)
        # group > 1: the prepare thread takes its items `group` at a time -- every item's own prepare() first (build, assemble), then ONE call
        # group_prepare([prepared objects]) for the lot (discretization.prefactor_many: their factorisations in the same launches), then they are handed to the
        # solve thread one by one.  The items' own prepare() must then leave the factorisation to group_prepare.
        self.group = 1 if self.strict else max(1, int(group))
        self.group_prepare = group_prepare if self.group > 1 else None
        # solvers > 1: that many solve threads take the prepared items in turn, so that the launches of item k+1's solve are already queued while the
        # host still waits for the residual norms of item k (every solve ends with a read-back the host has to look at).  The items of one pipeline
        # must then not share buffers they write (each solve thread's items run one after the other, those of different threads side by side).
        self.solvers = 1 if self.strict else max(1, int(solvers))
        self._threads = []

    def start(self, items):
        items = list(items)
        if not items:
            return
        if self.lookahead == 0:
            self._launch([threading.Thread(target=self._guard(self._serial), args=(items,), name='helm-dev%d' % self.device)])
            return
        ready = queue.Queue(maxsize=max(self.lookahead * self.group, self.solvers))
        # strict: item k+1 is prepared while item k is being solved and not before (a queue of one lets the prepare thread start on item k+2
        # as soon as item k+1 waits in it) -- for operators whose preparation is heavy on the GPU and in memory (3-D preconditioners)
        gate = threading.Semaphore(1) if self.strict else None
        tp = threading.Thread(target=self._guard(self._prepare_loop), args=(items, ready, gate), name='helm-prep%d' % self.device)
        if self.solvers == 1:
            tss = [threading.Thread(target=self._guard(self._solve_loop), args=(len(items), ready, gate), name='helm-solve%d' % self.device)]
        else:
            left = [len(items)]
            lock = threading.Lock()
            tss = [threading.Thread(target=self._guard(self._solve_shared), args=(left, lock, ready), name='helm-solve%d.%d' % (self.device, k)) for k in range(self.solvers)]
        self._launch([tp] + tss)

    def _launch(self, threads):
        'start the pipeline\'s threads; the process-wide BLAS limit (_QuietBlas) is held from now until the LAST of them has finished its items'
        self._alive = len(threads)
        self._alive_lock = threading.Lock()
        _QuietBlas.enter()
        for t in threads:
            t.daemon = True
            t.start()
        self._threads = threads

    def _guard(self, fn):
        def run(*args):
            try:
                fn(*args)
            finally:
                with self._alive_lock:
                    self._alive -= 1
                    last = self._alive == 0
                if last:
                    _QuietBlas.leave()
        return run

    def join(self):
        for t in self._threads:
            t.join()
        self._threads = []

    @staticmethod
    def _run_prepare(item):
        if item.prepare is not None:
            try:
                item._prepared = item.prepare()
            except BaseException as exc:       # delivered through the future by the solve thread
                item._error = exc

    @staticmethod
    def _run_solve(item):
        fut = item.future                  # (the consumer drops item.future once it has the result)
        if fut is None or not fut.set_running_or_notify_cancel():
            return
        if item._error is not None:
            fut.set_exception(item._error)
            return
        try:
            fut.set_result(item.solve(item._prepared))
        except BaseException as exc:
            fut.set_exception(exc)
        item._prepared = None

    def _serial(self, items):
        for item in items:
            self._run_prepare(item)
            self._run_solve(item)

    def _prepare_loop(self, items, ready, gate=None):
        if self.group > 1:
            # the first item goes by itself: the solve thread has nothing to do until something is ready, and a lone factorisation is ready in half the time of a
            # set of two (the pipeline fills 6-7 ms earlier; 2 % of a 20-item job)
            bounds = [0] + list(range(1 if len(items) > self.group else 0, len(items), self.group)) + [len(items)]
            for a, b in zip(bounds[:-1], bounds[1:]):
                part = items[a:b]
                if not part:
                    continue
                try:
                    for item in part:
                        self._run_prepare(item)
                    good = [item for item in part if item._error is None]
                    if self.group_prepare is not None and good:
                        self.group_prepare([item._prepared for item in good])
                except BaseException as exc:        # (an exception of the group's step belongs to every item of the group)
                    for item in part:
                        if item._error is None:
                            item._error = exc
                for item in part:
                    ready.put(item)
            return
        for item in items:
            try:
                if gate is not None:
                    gate.acquire()      # released when the solve of the previous item starts
                self._run_prepare(item)
            except BaseException as exc:    # (nothing above raises by design; but an item that never reaches the queue would leave the solve thread waiting for ever)
                item._error = exc
            ready.put(item)             # blocks while `lookahead` prepared items are waiting

    def _solve_loop(self, n, ready, gate=None):
        for _ in range(n):
            item = ready.get()
            if gate is not None:
                gate.release()
            self._run_solve(item)


    def _solve_shared(self, left, lock, ready):
        'one of several solve threads: takes the next prepared item as long as there is one left to take'
        while True:
            with lock:
                if left[0] <= 0:
                    return
                left[0] -= 1
            self._run_solve(ready.get())


def dispatch(workers, lookahead=1, strict=False, group=1, group_prepare=None):
    """Start one DevicePipeline per worker.  workers: [(device, [WorkItem, ...]), ...] (each list in the order it should run;
    the same device may appear twice: two pipelines then share that GPU).  Returns the pipelines (join() them, or just wait on
    the items' futures)."""
    if isinstance(workers, dict):
        workers = list(workers.items())
    pipes = []
    for dev, items in workers:
        p = DevicePipeline(dev, lookahead, strict, 1, group, group_prepare)
        p.start(items)
        pipes.append(p)
    return pipes


def pipelined(items, device=0, lookahead=1, strict=False, solvers=1, group=1, group_prepare=None):
    """Run `items` on one device with prepare-ahead and yield their results in order (exceptions surface where the
    failing item's result is consumed).  group / group_prepare: see DevicePipeline."""
    items = list(items)
    pipe = DevicePipeline(device, lookahead, strict, solvers, group, group_prepare)
    pipe.start(items)
    try:
        for item in items:
            res = item.future.result()
            item.future = None
            yield res
            del res
    finally:
        pipe.join()
