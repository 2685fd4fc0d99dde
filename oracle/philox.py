"""Philox4x32-10 in numpy -- TEST INFRASTRUCTURE (checker of the engine's sampler; never imported by the product).

The reference samples actions with torch's own generator (`dist.sample()`, agents/ppo.py:77); the engine draws its uniforms
from Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the counter-based
generator of the Random123 library) keyed by (seed, t*E + e).  This is the published algorithm restated; it is pinned by the
Random123 known-answer vectors below (tests/test_oracle_golden.py) and in turn pins the device implementation through
`mi_debug_philox` (tests/test_gpu_engine.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)

# Random123 kat_vectors, "philox4x32 10": counter (4 words), key (2 words) -> output (4 words)
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    """ctr (n,4) uint32, key (n,2) uint32 -> (n,4) uint32."""
    c = np.array(ctr, dtype=np.uint32).reshape(-1, 4).copy()
    k = np.array(key, dtype=np.uint32).reshape(-1, 2).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[:, 0].astype(np.uint64)
            p1 = M1 * c[:, 2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = np.stack([hi1 ^ c[:, 1] ^ k[:, 0], lo1, hi0 ^ c[:, 3] ^ k[:, 1], lo0], axis=1)
            k = np.stack([k[:, 0] + W0, k[:, 1] + W1], axis=1)
    return c


def uniform(seed, counter):
    """The sampler's uniform in [0,1) for a 64-bit seed and 64-bit counters: the top 24 bits of output word 0, times 2^-24
    (exact in fp32).  seed: int; counter: int array -> float32 array."""
    counter = np.asarray(counter, dtype=np.uint64).reshape(-1)
    ctr = np.zeros((counter.size, 4), np.uint32)
    ctr[:, 0] = (counter & np.uint64(0xffffffff)).astype(np.uint32)
    ctr[:, 1] = (counter >> np.uint64(32)).astype(np.uint32)
    key = np.empty((counter.size, 2), np.uint32)
    key[:, 0] = np.uint32(int(seed) & 0xffffffff)
    key[:, 1] = np.uint32((int(seed) >> 32) & 0xffffffff)
    out = philox4x32_10(ctr, key)
    return ((out[:, 0] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)
