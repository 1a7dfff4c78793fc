"""What the join's SIX- and TEN-byte tuples rest on (csrc/join.hip hash_a, p6_remainder, p10_key), checked on the CPU:

    hash_a(raw) = lowbias32(lo ^ hi * 0x9e3779b1)

  * lowbias32 -- xorshift 16, odd multiply, xorshift 15, odd multiply, xorshift 16 -- is a BIJECTION of 32-bit words: every step has an
    inverse (a right xorshift by s >= 11 is undone by two more of them, an odd multiplier has an inverse mod 2^32), built here and
    round-tripped;
  * so for a FIXED high word, lo -> hash_a(hi, lo) is a permutation of the low word: (hash_a, hi) determines the key, which is why a wide
    tuple may carry the hash's remaining bits + the high word instead of the key, and a narrow one (hi constant over the build range) the
    hash's remaining bits alone -- equal (partition, remainder, hi) means equal key, no false pair, no lost pair.

No GPU: this is arithmetic.  The device code is pinned to the same restatement by tests/test_gpu_join_internals.py::test_partition_invariants."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def lowbias32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def unxorshift(x, s):
    y = x.copy()                       # y = x ^ (x >> s)  =>  x = y ^ (y >> s) ^ (y >> 2s) ^ ...
    k = s
    while k < 32:
        y ^= x >> np.uint64(k)
        k += s
    return y


def lowbias32_inverse(h):
    inv1 = np.uint64(pow(0x846ca68b, -1, 1 << 32))
    inv0 = np.uint64(pow(0x7feb352d, -1, 1 << 32))
    x = unxorshift(h.astype(np.uint64), 16)
    x = (x * inv1) & M32
    x = unxorshift(x, 15)
    x = (x * inv0) & M32
    return unxorshift(x, 16)


def hash_a(raw):
    raw = raw.astype(np.uint64)
    fold = (raw & M32) ^ (((raw >> np.uint64(32)) * np.uint64(0x9e3779b1)) & M32)
    return lowbias32(fold)


def test_lowbias32_is_a_bijection_of_32_bit_words():
    rs = np.random.RandomState(1)
    x = np.concatenate([rs.randint(0, 1 << 32, size=2_000_000, dtype=np.uint64), np.arange(0, 1 << 20, dtype=np.uint64),
                        np.uint64(0xFFFFFFFF) - np.arange(0, 1 << 12, dtype=np.uint64)])
    h = lowbias32(x)
    assert h.max() <= 0xFFFFFFFF
    np.testing.assert_array_equal(lowbias32_inverse(h), x)
    # and onto: the inverse is a right inverse as well
    np.testing.assert_array_equal(lowbias32(lowbias32_inverse(x)), x)


def test_hash_and_high_word_determine_the_key():
    """ten-byte tuples: from (hash_a(key), hi) the key comes back -- for any high word"""
    rs = np.random.RandomState(2)
    raw = rs.randint(0, 1 << 62, size=1_000_000, dtype=np.int64).astype(np.uint64)
    raw[:4] = [0, 0xFFFFFFFF, 1 << 32, (1 << 62) - 1]
    hi = raw >> np.uint64(32)
    h = hash_a(raw)
    lo = lowbias32_inverse(h) ^ ((hi * np.uint64(0x9e3779b1)) & M32)
    np.testing.assert_array_equal((hi << np.uint64(32)) | lo, raw)


def test_narrow_keys_inside_one_2_32_window_need_no_high_word():
    """six-byte tuples: over a build range that does not straddle a 2^32 boundary the high word is a constant, the hash alone is
    injective; across a boundary it is not (two keys 2^32 apart with suitable low words share it) -- the library keeps 8-byte tuples there"""
    rs = np.random.RandomState(3)
    base = np.uint64(7) << np.uint64(32)
    keys = base + np.unique(rs.randint(0, 1 << 32, size=1_200_000, dtype=np.uint64))[:1_000_000]
    assert len(np.unique(hash_a(keys))) == len(keys)
    a = np.uint64(5) << np.uint64(32) | np.uint64(12345)
    hi_b = np.uint64(6)
    lo_b = (np.uint64(12345) ^ ((np.uint64(5) * np.uint64(0x9e3779b1)) & M32)) ^ ((hi_b * np.uint64(0x9e3779b1)) & M32)
    b = (hi_b << np.uint64(32)) | lo_b
    assert a != b and hash_a(np.array([a]))[0] == hash_a(np.array([b]))[0]
