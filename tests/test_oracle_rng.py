"""The oracle's numpy.random restatement against the installed numpy (2.2.6 = the reference's pin,
docs/environment_full.yml:154).  Bit-exact: PCG64 raw stream, SeedSequence seeding, random(),
uniform(), integers()/choice(n) (buffered Lemire), choice(n, p=), permutation([1,2]), round(.,2)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

SEEDS = [0, 1, 2, 3, 42, 99, 12345, 2**31 - 1, 2**32 - 1, 2**32, 2**32 + 98, 2**40 + 7, 2**63 + 11, 2**64 - 1]


def _state_words(gen):
    st = gen.bit_generator.state
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    m = (1 << 64) - 1
    return (s >> 64, s & m, inc >> 64, inc & m, int(st["has_uint32"]), int(st["uinteger"]))


def _words(g):
    return (g.state_hi, g.state_lo, g.inc_hi, g.inc_lo, g.has_uint32, g.uinteger)


@pytest.mark.parametrize("seed", SEEDS)
def test_seeding_and_raw_stream(seed):
    L = O.lib()
    g = O.Pcg64()
    L.orc_pcg64_seed(C.byref(g), seed)
    ref = np.random.default_rng(seed)
    assert _words(g) == _state_words(ref)
    raw = ref.bit_generator.random_raw(64)
    mine = [L.orc_pcg64_next64(C.byref(g)) for _ in range(64)]
    assert mine == [int(x) for x in raw]
    assert _words(g) == _state_words(ref)


@pytest.mark.parametrize("seed", SEEDS[:8])
def test_random_uniform_integers_interleaved(seed):
    L = O.lib()
    g = O.Pcg64()
    L.orc_pcg64_seed(C.byref(g), seed)
    ref = np.random.default_rng(seed)
    script = np.random.default_rng(1000 + seed % 1000)
    for _ in range(400):
        op = int(script.integers(0, 4))
        if op == 0:
            assert L.orc_pcg64_random(C.byref(g)) == ref.random()
        elif op == 1:
            n = float(script.choice([0.05, 0.3, 1e-3]))
            assert L.orc_pcg64_uniform(C.byref(g), -n, n) == ref.uniform(-n, n)
        elif op == 2:
            lo, hi = 60, 81
            assert L.orc_pcg64_integers(C.byref(g), lo, hi) == int(ref.integers(lo, hi))
        else:
            k = int(script.integers(1, 12))
            arr = np.sort(script.choice(11, size=k, replace=False))
            got = int(arr[L.orc_pcg64_integers(C.byref(g), 0, k)])
            assert got == int(ref.choice(arr))
        assert _words(g) == _state_words(ref)


def test_uniform_vector_equals_scalar_sequence():
    # update_accuracy draws uniform(-n, n, 4) as one call (env_super.py:508)
    L = O.lib()
    for seed in (4, 7, 103):
        g = O.Pcg64()
        L.orc_pcg64_seed(C.byref(g), seed)
        ref = np.random.default_rng(seed)
        for _ in range(50):
            vec = ref.uniform(-0.05, 0.05, 4)
            mine = [L.orc_pcg64_uniform(C.byref(g), -0.05, 0.05) for _ in range(4)]
            assert mine == vec.tolist()


@pytest.mark.parametrize("seed", [5, 99, 102])
def test_choice_with_probabilities(seed):
    # sort_material's draw: p = leftover / leftover.sum() (env_super.py:562-563)
    L = O.lib()
    g = O.Pcg64()
    L.orc_pcg64_seed(C.byref(g), seed)
    ref = np.random.default_rng(seed)
    script = np.random.default_rng(seed + 17)
    for _ in range(3000):
        left = script.integers(0, 45, size=4)
        if script.random() < 0.3:
            left[script.integers(0, 4)] = 0
        if left.sum() == 0:
            left[1] = 3
        p = left / left.sum()
        pp = (C.c_double * 4)(*p.tolist())
        assert L.orc_pcg64_choice_p(C.byref(g), pp, 4) == int(ref.choice(4, p=p))
    assert _words(g) == _state_words(ref)


def test_permutation_first_pattern():
    L = O.lib()
    for seed in list(range(0, 600)) + SEEDS:
        exp = int(np.random.default_rng(seed).permutation([1, 2])[0])
        assert L.orc_permutation12_first(seed) == exp, seed


def test_round2_and_rint():
    L = O.lib()
    for total in range(1, 400):
        for true in range(0, total + 1, max(1, total // 37)):
            x = np.float64(true) / np.float64(total)
            assert L.orc_round2(float(x)) == float(round(x, 2))
            d = round(x, 2) - 0.9
            assert L.orc_round2(float(d)) == float(round(d, 2))
    for n in range(0, 120):
        for acc in (0.75, 1.0, 0.7, 0.7123456, 0.8, 0.5, 0.25):
            v = np.int64(n) * acc
            assert L.orc_rint_i64(float(v)) == int(round(v))
