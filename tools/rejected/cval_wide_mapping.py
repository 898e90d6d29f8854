#!/usr/bin/env python3
"""Index simulation of r04_constraint_rows_wide_stores.patch for one wave: the LDS words [output row][lane], the nb // 4 wide stores
(quarter-wave q of store j = row 4 j + q, 16 bytes = four consecutive envs per lane) and the narrow stores of the remaining rows must
write every (row, env) word of the wave's block exactly once, at c_values[row][env].  Checked for the shipped configs' row tables
(Quadrotor2D 16 rows, CartPole 10, Quadrotor3D 32) with random output-row permutations."""
import numpy as np


def check(nb, perm, N=65536, wave_base=64 * 37):
    lds = np.full(nb * 64, np.nan)
    for lane in range(64):
        for r in range(nb):
            lds[perm[r] * 64 + lane] = 1000 * perm[r] + (wave_base + lane)
    mem = {}

    def put(addr, v):
        assert addr not in mem, addr
        mem[addr] = v
    for lane in range(64):
        q, e4 = lane >> 4, (lane & 15) * 4
        wave_off = (wave_base + lane) * 4 - lane * 4
        for j in range(nb // 4):
            row = 4 * j + q
            for k in range(4):
                put(wave_off + e4 * 4 + q * N * 4 + 4 * j * N * 4 + 4 * k, lds[row * 64 + e4 + k])
        for r in range(nb):                                  # narrow remainder: c_out.store(cv[r], o * stride) at the lane's own env
            o = perm[r]
            if o >= nb // 4 * 4:
                put((wave_base + lane) * 4 + o * N * 4, 1000 * o + (wave_base + lane))
    assert len(mem) == nb * 64
    for a, v in mem.items():
        assert v == 1000 * (a // (N * 4)) + (a % (N * 4)) // 4, (a, v)


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for nb in (16, 10, 32, 4, 7):
        for _ in range(5):
            check(nb, [int(v) for v in rng.permutation(nb)])
    check(16, [0, 6, 1, 7, 2, 8, 3, 9, 4, 10, 5, 11, 12, 14, 13, 15])          # quadrotor_2D_track's table
    print('ok: every (row, env) word of the wave once, at c_values[row][env]')
