"""Persistent host-side build workers (map generation, scene placement: reset-time work, all numpy).

A process that has initialised the GPU must neither fork nor exec (the children would inherit a HIP runtime they cannot
use; on the GPU pool it can take the machine down).  So the workers are plain child interpreters started BEFORE the
first GPU call -- `start()` refuses afterwards -- and kept for the life of the process: every later build (a
`reset(seed=...)` with new scenario seeds, the next engine of a test session) is a message to them, never a new
process.  The workers import only the numpy side of the package (mapgen / scene / scenario), never torch, and are
started with the GPU hidden from them.

Protocol: length-prefixed pickles over the worker's stdin / stdout (its prints go to stderr).  A job names a module-level
function (`module`, `name`) and carries a chunk of arguments; the reply is the list of results or the formatted exception.
`map()` keeps the order of the jobs.  Optional memo (`cache=True`): the pickled result of a job is kept by its key, so
that building the same scenario seed again (sub-batches of the same envs, the CPU baseline's copy) costs one unpickle and
yields a fresh object every time.
"""
import atexit
import hashlib
import importlib
import os
import pickle
import struct
import subprocess
import sys
import threading
import traceback

_POOL = None
_LOCK = threading.Lock()
_MEMO = {}          # job key -> pickled result
_MEMO_BYTES = [0]
_MEMO_LIMIT = int(os.environ.get("MD_BUILD_MEMO_MB", "4096")) << 20


def gpu_initialised():
    """True once this process has (or may have) a HIP context; never imports torch to find out."""
    if os.environ.get("HSA_TOOLS_LIB") or "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_ENABLED"):
        return True      # under rocprofv3 the profiler's library makes the context before main() runs
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


def default_workers():
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:      # the GPU boxes give a 16-CPU quota on a 256-thread host
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("MD_BUILD_WORKERS", "0")) or max(1, min(n, 32))


def _send(fh, obj):
    b = pickle.dumps(obj, protocol=4)
    fh.write(struct.pack("<Q", len(b)))
    fh.write(b)
    fh.flush()


def _recv(fh):
    h = fh.read(8)
    if len(h) < 8:
        raise EOFError("build worker closed its pipe")
    n = struct.unpack("<Q", h)[0]
    b = fh.read(n)
    if len(b) < n:
        raise EOFError("build worker closed its pipe")
    return pickle.loads(b)


class BuildPool:
    def __init__(self, workers):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        env["HIP_VISIBLE_DEVICES"] = ""            # the workers have no business on the GPU
        env["MD_BUILD_WORKER"] = "1"
        env.setdefault("OMP_NUM_THREADS", "1")
        env.setdefault("OPENBLAS_NUM_THREADS", "1")
        self.procs = [subprocess.Popen([sys.executable, "-u", "-m", "metadrive_ped_amd.hostpool"], stdin=subprocess.PIPE,
                                       stdout=subprocess.PIPE, env=env, close_fds=True) for _ in range(workers)]
        self.busy = threading.Lock()

    def alive(self):
        return all(p.poll() is None for p in self.procs)

    def map(self, fn, jobs, chunk=0, sticky=False):
        """[fn(j) for j in jobs] on the workers; fn is a module-level function.  sticky: job i always goes to worker
        (i // chunk) % workers with a chunk size that depends on len(jobs) only -- the same job list meets the same workers
        again, whose per-process caches (engine._MAP_CACHE) then hit; otherwise the workers take chunks as they get free."""
        jobs = list(jobs)
        n = len(jobs)
        if n == 0:
            return []
        chunk = chunk or max(1, min(32, n // (len(self.procs) * 4) or 1))
        starts = list(range(0, n, chunk))
        out = [None] * n
        errors = []
        cursor = [0]
        take = threading.Lock()
        mine = {id(p): [a for k, a in enumerate(starts) if k % len(self.procs) == w] for w, p in enumerate(self.procs)}

        def drive(p):
            while not errors:
                with take:
                    if sticky:
                        if not mine[id(p)]:
                            return
                        a = mine[id(p)].pop(0)
                    else:
                        if cursor[0] >= len(starts):
                            return
                        a = starts[cursor[0]]
                        cursor[0] += 1
                try:
                    _send(p.stdin, (fn.__module__, fn.__name__, jobs[a:a + chunk]))
                    ok, res = _recv(p.stdout)
                except Exception as ex:       # a dead worker
                    errors.append("build worker failed: %r" % (ex, ))
                    return
                if not ok:
                    errors.append(res)
                    return
                out[a:a + len(res)] = res

        with self.busy:
            threads = [threading.Thread(target=drive, args=(p, ), daemon=True) for p in self.procs]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errors:
            err = errors[0]
            if isinstance(err, tuple):
                # what the build code raises for a config it refuses (ValueError: too many movers for the capacity ...,
                # NotImplementedError) reaches the caller as that type, as it does when the build runs in this process
                kind, text, trace = err
                if kind in _PASS_THROUGH:
                    raise _PASS_THROUGH[kind](text)
                err = trace
            raise RuntimeError("host build failed in a worker:\n" + err)
        return out

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()


def start(workers=0):
    """Start the workers (idempotent).  Must happen before the process touches the GPU; afterwards it raises."""
    global _POOL
    with _LOCK:
        if _POOL is not None and _POOL.alive():
            return _POOL
        if gpu_initialised():
            raise RuntimeError("metadrive_ped_amd.hostpool.start(): this process has already initialised the GPU; start the "
                               "build workers first (import metadrive_ped_amd.hostpool; hostpool.start())")
        _POOL = BuildPool(workers or default_workers())
        atexit.register(stop)
        return _POOL


def stop():
    global _POOL
    with _LOCK:
        if _POOL is not None:
            _POOL.close()
            _POOL = None


def get(auto_start=True):
    """The running pool; started now if allowed (no GPU context yet, not inside a worker), else None = build serially."""
    if os.environ.get("MD_BUILD_WORKER") == "1" or os.environ.get("MD_BUILD_WORKERS") == "1":
        return None
    if _POOL is not None and _POOL.alive():
        return _POOL
    if not auto_start or gpu_initialised():
        return None
    return start()


_PASS_THROUGH = {"ValueError": ValueError, "NotImplementedError": NotImplementedError}


def build_all(fn, jobs, workers=0, cache=False, min_parallel=16, sticky=False):
    """[fn(j) for j in jobs]: on the persistent workers when there are enough jobs and the pool is (or may be) running,
    serially otherwise.  cache=True memoises pickled results by the hash of the pickled job; sticky: see Pool.map."""
    jobs = list(jobs)
    if workers == 1:
        return [fn(j) for j in jobs]
    keys = None
    out = [None] * len(jobs)
    todo = list(range(len(jobs)))
    if cache:
        keys = [(fn.__module__, fn.__name__, hashlib.sha1(pickle.dumps(j, protocol=4)).digest()) for j in jobs]
        todo = []
        for i, k in enumerate(keys):
            b = _MEMO.get(k)
            if b is None:
                todo.append(i)
            else:
                out[i] = pickle.loads(b)
    pool = get() if len(todo) >= min_parallel else None
    res = pool.map(fn, [jobs[i] for i in todo], sticky=sticky) if pool is not None else [fn(jobs[i]) for i in todo]
    for i, r in zip(todo, res):
        out[i] = r
        if cache and _MEMO_BYTES[0] < _MEMO_LIMIT:
            b = pickle.dumps(r, protocol=4)
            _MEMO[keys[i]] = b
            _MEMO_BYTES[0] += len(b)
    return out


def worker_pid(_job=None):
    """The process a job runs in (tests: sticky routing)."""
    return os.getpid()


def clear_memo():
    _MEMO.clear()
    _MEMO_BYTES[0] = 0


def _worker_main():
    inp = sys.stdin.buffer
    out = os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)                          # anything the build code prints goes to stderr, not into the protocol
    sys.stdout = sys.stderr
    fns = {}
    while True:
        try:
            mod, name, chunk = _recv(inp)
        except EOFError:
            return
        try:
            f = fns.get((mod, name))
            if f is None:
                f = fns[(mod, name)] = getattr(importlib.import_module(mod), name)
            _send(out, (True, [f(j) for j in chunk]))
        except Exception as ex:
            _send(out, (False, (type(ex).__name__, str(ex), traceback.format_exc())))


if __name__ == "__main__":
    _worker_main()
