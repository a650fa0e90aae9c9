"""HE Wrapper/Utils.cs mirror: ProcessInEnv / ParallelProcessInEnv.  CPU: on the plaintext factory.  GPU: many host threads issuing
evaluator calls against one encrypted factory (what the reference does from Defaults.ThreadCount threads)."""
import numpy as np
import pytest

from cryptonets_amd import utils
from cryptonets_amd.hewrapper import EVectorFormat
from cryptonets_amd.raw import RawFactory


def test_process_in_env_and_parallel_on_raw(capsys):
    f = RawFactory(64)
    v = f.GetEncryptedVector(np.arange(1.0, 5.0), EVectorFormat.dense, 1)
    assert utils.ProcessInEnv(lambda env: v.SumAllSlots(env).Decrypt(env)[0], f) == 10.0
    seen, tasks = [None] * 50, set()

    def item(env, task, k):
        seen[k] = v.Multiply(k, env).Decrypt(env)[1]
        tasks.add(task)
    utils.ParallelProcessInEnv(50, item, f)
    assert seen == [2.0 * k for k in range(50)] and tasks <= set(range(utils.Defaults.ThreadCount))
    one = []
    utils.ParallelProcessInEnv(1, lambda env, task, k: one.append((task, k)), f)
    utils.ParallelProcessInEnv(0, lambda env, task, k: one.append("never"), f)
    master = f.AllocateComputationEnv()
    utils.ParallelProcessInEnv(1, lambda env, task, k: one.append(env is master), masterEnv=master)
    assert one == [(0, 0), True]
    with pytest.raises(ValueError, match="item 7"):
        def bad(env, task, k):
            if k == 7:
                raise ValueError("item 7")
        utils.ParallelProcessInEnv(20, bad, f)
    utils.Time("nothing", lambda: None)
    assert "Time for nothing:" in capsys.readouterr().out


@pytest.mark.gpu
def test_parallel_process_in_env_on_the_device():
    """32 items from up to 8 threads on one factory (two plaintext primes): per item encrypt -> square -> rotate-and-add -> decrypt"""
    from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
    Factory = EncryptedSealBfvFactory([65537, 114689], 4096)
    old, utils.Defaults.ThreadCount = utils.Defaults.ThreadCount, 8
    out = [None] * 32
    try:
        def item(env, task, k):
            v = Factory.GetEncryptedVector(np.arange(1.0, 9.0) + k, EVectorFormat.dense, 1)
            out[k] = v.DotProduct(v, env).Decrypt(env)[0]
            v.Dispose()
        utils.ParallelProcessInEnv(32, item, Factory)
    finally:
        utils.Defaults.ThreadCount = old
    assert out == [float(np.sum((np.arange(1.0, 9.0) + k) ** 2)) for k in range(32)]
