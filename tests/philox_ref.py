"""numpy restatement of the eta-noise generator of said_amd/csrc/sched_math.h — TEST INFRASTRUCTURE.

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
library's philox4x32 with 10 rounds), pinned below against Random123's published known-answer vectors, followed by the
engine's mapping to one standard normal per (seed, step, element): counter = (element, step, 0, 0), key = (seed low word,
seed high word), u1 = ((r0 >> 8) + 1) / 2**24, u2 = (r1 >> 8) / 2**24, z = sqrt(-2 ln u1) cos(2 pi u2).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)

# Random123 kat_vectors, "philox4x32 10": (counter[4], key[2]) -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (broadcastable); returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    k0, k1 = np.asarray(k0, dtype=np.uint32), np.asarray(k1, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return c0, c1, c2, c3


def normals(seed: int, step0: int, nsteps: int, n_per_step: int) -> np.ndarray:
    """(nsteps, n_per_step) float64 standard normals of steps step0 .. step0 + nsteps - 1."""
    elem = np.arange(n_per_step, dtype=np.uint32)[None, :]
    step = (step0 + np.arange(nsteps, dtype=np.uint32))[:, None]
    z = np.zeros_like(elem + step)
    r0, r1, _, _ = philox4x32_10(elem + z, step + z, z, z, np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    u1 = ((r0 >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
    u2 = (r1 >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
