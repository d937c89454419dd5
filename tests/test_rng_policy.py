"""The random-number policy: the recurrence is libc's drand48, restarted per path."""
import ctypes

import numpy as np


def test_recurrence_is_libc_drand48(po):
    libc = ctypes.CDLL("libc.so.6")
    libc.drand48.restype = ctypes.c_double
    libc.seed48.argtypes = [ctypes.POINTER(ctypes.c_ushort)]
    for state0 in (0x1234ABCD330E, 0, 0xFFFFFFFFFFFF, po.lib().orc_path_state(7, 12345)):
        seed = (ctypes.c_ushort * 3)(state0 & 0xFFFF, (state0 >> 16) & 0xFFFF, (state0 >> 32) & 0xFFFF)
        libc.seed48(seed)
        st = ctypes.c_uint64(state0)
        for _ in range(50):
            assert po.lib().orc_rng_draw(ctypes.byref(st)) == libc.drand48()


def test_path_states_are_spread(po):
    s = np.array([po.lib().orc_path_state(1, i) for i in range(4096)], np.uint64)
    assert len(set(s.tolist())) == 4096 and (s < (1 << 48)).all()
    first = np.array([(int(x) * 0x5DEECE66D + 0xB) & ((1 << 48) - 1) for x in s], np.float64) / 2.0 ** 48
    assert abs(first.mean() - 0.5) < 0.02 and abs(np.corrcoef(first[:-1], first[1:])[0, 1]) < 0.06
    assert po.lib().orc_path_state(1, 5) != po.lib().orc_path_state(2, 5)
