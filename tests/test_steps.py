"""Step arithmetic of the wire protocol (SS/engine/ipc_service.cu:60-132,213-253): the library's
IPCEnv (host-only calls, no GPU) against the oracle, including uneven partitions."""
import ctypes
import os

import numpy as np
import pytest

from oracle import ffi


@pytest.fixture()
def hiplib():
    os.environ["LEGION_IPC_LOCAL"] = "1"          # keep the slab in process memory, no /dev/shm entry
    from legion_amd import lib
    return lib.load()


@pytest.mark.parametrize("train,valid,test,bs,epoch", [
    ([196615], [39323], [2213091], 1024, 2),
    ([1670413, 1670415, 1670412, 1670414, 1670413, 1670415, 1670411, 1670411], [12500] * 8, [12499] * 8, 8000, 10),
    ([700, 650, 900], [37, 5, 90], [5, 9, 1], 64, 3),
    ([5000, 5000], [513, 511], [1024, 1], 512, 1),
])
def test_steps_match_oracle(hiplib, oracle, train, valid, test, bs, epoch):
    P = len(train)
    arr = lambda v: (ctypes.c_int32 * P)(*v)
    env = hiplib.NewIPCEnv(P)
    hiplib.legion_ipc_coordinate(env, P, arr(train), arr(valid), arr(test), bs, epoch)
    st = ffi.Steps()
    oracle.lgo_coordinate(ctypes.byref(st), P, arr(train), arr(valid), arr(test), bs, epoch)
    assert hiplib.legion_ipc_train_step(env) == st.train_step == (min(train) - 1) // bs
    assert hiplib.legion_ipc_max_step(env) == oracle.lgo_max_step(ctypes.byref(st))
    total = hiplib.legion_ipc_max_step(env)
    for g in list(range(min(total, 300))) + [total - 1]:
        assert hiplib.legion_ipc_current_mode(env, g) == oracle.lgo_current_mode(ctypes.byref(st), g)
        assert hiplib.legion_ipc_local_batch_id(env, g) == oracle.lgo_local_batch_id(ctypes.byref(st), g)
    for d in range(P):
        for m in range(3):
            assert hiplib.legion_ipc_current_batchsize(env, d, m) == oracle.lgo_current_batchsize(ctypes.byref(st), d, m)
    hiplib.legion_ipc_finalize(env)


def test_schedule_shape(oracle):
    st = ffi.Steps()
    one = lambda v: (ctypes.c_int32 * 1)(v)
    oracle.lgo_coordinate(ctypes.byref(st), 1, one(2049), one(1000), one(600), 1024, 2)
    assert (st.train_step, st.valid_step, st.test_step) == (2, 2, 2)
    modes = [oracle.lgo_current_mode(ctypes.byref(st), g) for g in range(oracle.lgo_max_step(ctypes.byref(st)))]
    assert modes == [0, 0, 1, 1, 0, 0, 1, 1, 2, 2]
    assert st.valid_bs[0] == 500 and st.test_bs[0] == 300
