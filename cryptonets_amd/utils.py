"""HE Wrapper/Utils.cs: run user code inside a computation environment - once, or over `count` items from `Defaults.ThreadCount`
threads that pull item numbers from a shared counter (:46-88).  On the device the heavy layers are single batched calls, so nothing in
`layers.py` needs the thread pool; it is here for user code written against the reference (`Utils.ProcessInEnv(env => ..., Factory)`,
`Basic Example/Program.cs:35`).  libcnhip serialises callers per context, so concurrent lambdas are safe."""
import os
import threading
import time


class Defaults:
    """HE Wrapper/Defaults.cs (`RawFactory` lives in cryptonets_amd.raw.Defaults)"""
    ThreadCount = os.cpu_count() or 1


def ProcessInEnv(fn, factory):
    """:16-37: allocate an environment, run `fn(env)`, free the environment; returns what fn returns"""
    env = factory.AllocateComputationEnv()
    try:
        return fn(env)
    finally:
        factory.FreeComputationEnv(env)


def ParallelProcessInEnv(count, fn, factory=None, masterEnv=None):
    """:39-88: `fn(env, taskIndex, k)` for k in 0..count-1.  One item (or none): on the caller's thread in `masterEnv` (or a fresh
    environment); otherwise min(ThreadCount, count) threads, each with its own environment, take the next k from a shared counter."""
    if factory is None:
        factory = masterEnv.ParentFactory
    if count < 2:
        if masterEnv is not None:
            for k in range(count):
                fn(masterEnv, 0, k)
        else:
            ProcessInEnv(lambda env: [fn(env, 0, k) for k in range(count)], factory)
        return
    lock, state, errors = threading.Lock(), {"next": -1}, []

    def worker(task_index):
        env = factory.AllocateComputationEnv()
        try:
            while True:
                with lock:
                    state["next"] += 1
                    k = state["next"]
                if k >= count or errors:
                    break
                fn(env, task_index, k)
        except BaseException as e:                                # Task.WaitAll rethrows: so do we, after joining
            errors.append(e)
        finally:
            factory.FreeComputationEnv(env)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(min(Defaults.ThreadCount, count))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def Time(name, fn):
    """:90-96"""
    start = time.perf_counter()
    fn()
    print("Time for %s: %s" % (name, 1e3 * (time.perf_counter() - start)))
