"""The harness models (tests/harness_models: MT19937, xoshiro256**, LFSRs) against independent authorities: CPython's own generator, the
published first outputs of xoshiro256** for the state (1, 2, 3, 4), and the period of a primitive polynomial -- on concrete ints; the symbolic
side (BitVec / PackedBitVec words) is pinned by the golden equation fingerprints captured from the reference's Python layer."""
import random

from tests.harness_models import MT19937, FibonacciLFSR, GaloisLFSR, Xoshiro256starstar


def test_mt19937_follows_cpython_random():
    for seed in (0, 3142, 2 ** 40 + 7):
        ref = random.Random(seed)
        st = ref.getstate()[1]
        gen = MT19937(st[:-1])
        gen.mti = st[-1]
        for k in (32, 1, 9, 17, 31, 33, 64, 137, 1337, 32):
            for _ in range(40):                        # (624 words are used up several times: the recurrence runs, too)
                assert gen.getrandbits(k) == ref.getrandbits(k), (seed, k)
        follow = gen.to_python_random()
        assert [follow.getrandbits(32) for _ in range(700)] == [ref.getrandbits(32) for _ in range(700)]
    assert MT19937([0] * 624).getrandbits(0) == 0


def test_xoshiro256starstar_known_outputs_and_inverse_scrambler():
    gen = Xoshiro256starstar([1, 2, 3, 4])
    # first outputs of the reference C code (xoshiro256starstar.c) for s = {1, 2, 3, 4}
    assert [gen() for _ in range(4)] == [11520, 0, 1509978240, 1215971899390074240]
    rng = random.Random(5)
    for _ in range(200):
        x = rng.getrandbits(64)
        assert Xoshiro256starstar.unscramble(Xoshiro256starstar.scramble(x)) == x
        assert Xoshiro256starstar.scramble(Xoshiro256starstar.unscramble(x)) == x
    g = Xoshiro256starstar([rng.getrandbits(64) for _ in range(4)])
    s1 = g.s[1]
    assert g.step() == s1                              # step() hands back the PRE-update s1: what the scrambler is applied to


def test_lfsr_forms_have_the_period_of_their_polynomial():
    # x^4 + x + 1 is primitive: both forms walk all 15 non-zero states; taps as masks of a right-shifting register
    for cls, mask in ((GaloisLFSR, 0b1100), (FibonacciLFSR, 0b0011)):
        reg = cls(4, mask, 1)
        seen, bits = [], []
        for _ in range(15):
            seen.append(reg.state)
            bits.append(reg())
        assert len(set(seen)) == 15 and 0 not in seen and reg.state == seen[0], (cls.__name__, seen)
        assert sum(bits) == 8                          # an m-sequence of length 15 has eight ones
