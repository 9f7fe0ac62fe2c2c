"""CPU model of the sorting network of csrc/antq_k_sortsearch.h (the sorted-row clip search).

The kernel sorts 4096 keys with 256 threads x 16 registers: every compare-exchange runs on registers, the index bits that ARE
register bits change with the layout ([lo, lo + 4) of the 12 index bits), a merge moves between layouts through LDS at dword
address i + (i >> 4), and the first step of every merge (partner i ^ (2^s - 1)) is folded into the transposing READ: the upper
half block comes in mirrored, so no comparator needs a direction.  This file restates exactly that index arithmetic in numpy
(sort_lay_lo / sort_lay_base / sort_lay_off / sort_load's mirror rule / sort_wg's phase structure: sizes <= 16 in registers,
<= 1024 inside a wavefront with 10 index bits, 2048 and 4096 across the workgroup) and checks that it sorts, how many
transposes it takes, and that every layout's LDS accesses stay close to conflict-free on 32 banks.  It guards the
DESIGN's description of the algorithm; the HIP code itself is checked on the GPU (tests/test_gpu_sort_r6.py)."""
import numpy as np
import pytest

R = 4


def lay_lo(b, bt):
    return min((b // R) * R, bt - R)


def idx(t, r, lo):
    return ((t >> lo) << (lo + R)) | (r << lo) | (t & ((1 << lo) - 1))


def phys(i):
    return i + (i >> R)


def bank_cost(addrs):
    """LDS cycles of one ds_read_b32 / ds_write_b32 of a wavefront: two groups of 32 lanes, 32 banks of 4 bytes."""
    c = 0
    for g in range(0, len(addrs), 32):
        a = addrs[g:g + 32]
        c += max(len(set(x for x in a if x % 32 == b)) for b in range(32))
    return c


class Model:
    def __init__(self, B, seed):
        self.B, self.K, self.NT, self.EPT = B, 1 << B, 1 << (B - R), 1 << R
        rng = np.random.default_rng(seed)
        self.keys = rng.integers(0, 1 << 32, self.K, dtype=np.uint64).astype(np.uint32)
        self.keys[rng.integers(0, self.K, self.K // 8)] = 0xFFFFFFFF            # sentinels (literal elements / the row's tail)
        self.lds = np.zeros(phys(self.K - 1) + 1, dtype=np.uint32)
        self.t = np.arange(self.NT)
        self.regs = np.stack([self.keys[idx(self.t, r, 0)] for r in range(self.EPT)], 1)
        self.transposes = 0
        self.cost = self.ideal = 0

    def _acc(self, a):
        for w in range(0, self.NT, 64):
            self.cost += bank_cost(list(a[w:w + 64]))
            self.ideal += 2

    def ce(self, r, r2):
        a, b = self.regs[:, r].copy(), self.regs[:, r2].copy()
        self.regs[:, r], self.regs[:, r2] = np.minimum(a, b), np.maximum(a, b)

    def ce_bit(self, bit):
        for r in range(self.EPT):
            if not (r >> bit) & 1:
                self.ce(r, r | (1 << bit))

    def store(self, lo, base, tl):
        for r in range(self.EPT):
            a = base + phys(idx(tl, r, lo))
            self.lds[a] = self.regs[:, r]
            self._acc(a)

    def load(self, lo, base, tl, s):
        for r in range(self.EPT):
            if s and (r >> (s - 1 - lo)) & 1:                       # the upper half block of the size-2^s merge: mirrored
                a = base + phys(idx(tl ^ ((1 << lo) - 1), r ^ ((1 << (s - 1 - lo)) - 1), lo))
            else:
                a = base + phys(idx(tl, r, lo))
            self.regs[:, r] = self.lds[a]
            self._acc(a)

    def steps(self, bt, s, base, tl):
        b, cur, first = s - 1, 0, True
        while b >= 0:
            lo = lay_lo(b, bt)
            self.store(cur, base, tl)
            self.load(lo, base, tl, s if first else 0)
            self.transposes += 1
            for bit in range(b - lo, -1, -1):
                self.ce_bit(bit)
            b, cur, first = lo - 1, lo, False

    def sort(self, wave_bits=None):
        for s in range(1, R + 1):                                   # sizes 2 .. 16: registers only
            for r in range(self.EPT):
                if not (r >> (s - 1)) & 1:
                    self.ce(r, r ^ ((1 << s) - 1))
            for bit in range(s - 2, -1, -1):
                self.ce_bit(bit)
        bw = self.B if wave_bits is None else wave_bits
        if bw < self.B:                                             # sizes 32 .. 2^bw inside each wavefront's own block
            wave, lane = self.t >> (bw - R), self.t & ((1 << (bw - R)) - 1)
            base = wave * ((1 << bw) + (1 << (bw - R)))
            for s in range(R + 1, bw + 1):
                self.steps(bw, s, base, lane)
        for s in range(max(R, bw if bw < self.B else R) + 1, self.B + 1):
            self.steps(self.B, s, 0, self.t)
        out = np.zeros(self.K, dtype=np.uint32)
        for r in range(self.EPT):
            out[idx(self.t, r, 0)] = self.regs[:, r]
        return out


@pytest.mark.parametrize("seed", range(3))
def test_network_sorts_with_in_wave_phases(seed):
    m = Model(12, seed)
    out = m.sort(wave_bits=10)
    assert np.array_equal(out, np.sort(m.keys))
    assert m.transposes == 14 + 6                     # 14 inside the wavefronts (no workgroup barrier), 6 across the workgroup
    assert m.cost <= 1.35 * m.ideal                   # address i + (i >> 4): every layout within 35 % of conflict-free


def test_network_sorts_without_the_in_wave_split_and_other_sizes():
    m = Model(12, 7)
    assert np.array_equal(m.sort(), np.sort(m.keys)) and m.transposes == 20
    for B in (8, 10, 11):
        m = Model(B, B)
        assert np.array_equal(m.sort(), np.sort(m.keys))


def test_key_order_equals_float_order():
    """sort_key: the float's bits as an unsigned integer of the same order, -0 folded into +0; sort_unkey inverts it."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 4000).astype(np.float32),
                        np.float32([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, -3.4e38])])
    u = np.where(x == 0, np.uint32(0), x.view(np.uint32))
    key = np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint32)
    order_k, order_x = np.argsort(key, kind="stable"), np.argsort(np.where(x == 0, np.float32(0.0), x), kind="stable")
    assert np.array_equal(x[order_k] + np.float32(0.0), x[order_x] + np.float32(0.0))
    back = np.where(key & 0x80000000, key ^ np.uint32(0x80000000), ~key).astype(np.uint32).view(np.float32)
    assert np.array_equal(back, np.where(x == 0, np.float32(0.0), x))
    assert key.max() < 0xFFFFFFFF                     # the sentinel sorts behind every float, +Inf included
