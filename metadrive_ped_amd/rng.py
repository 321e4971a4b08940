"""Seeded numpy RandomState factory with the reference's seed hashing.

Restates metadrive/utils/random_utils.py:14-105 (get_np_random / hash_seed / create_seed): the
reference never seeds RandomState with the raw integer; it SHA-512-hashes str(seed), keeps the first
8 bytes as a little-endian bigint and seeds with its 32-bit limbs.  Map topology, block parameters,
spawn lanes and traffic all hang off streams built this way, so "seed => world" parity starts here.
Known answer: get_np_random(1010).randint(0, 65536) == 30146.
"""
import hashlib
import struct

import numpy as np


def _bigint_from_bytes(b):
    pad = 4 - len(b) % 4
    b = b + b"\0" * pad
    n = len(b) // 4
    vals = struct.unpack("{}I".format(n), b)
    acc = 0
    for i, v in enumerate(vals):
        acc += (1 << (32 * i)) * v
    return acc


def hash_seed(seed, max_bytes=8):
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def _int_list_from_bigint(x):
    if x == 0:
        return [0]
    out = []
    while x > 0:
        x, mod = divmod(x, 1 << 32)
        out.append(mod)
    return out


def get_np_random(seed):
    """RandomState seeded exactly like the reference does for a non-negative int seed."""
    if not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError("Seed must be a non-negative integer, not {!r}".format(seed))
    seed = int(seed) % (1 << 64)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng


class Randomizable:
    """metadrive/base_class/randomizable.py:4-21"""
    MAX_RAND_INT = 65536

    def __init__(self, seed):
        self.seed(seed)

    def seed(self, seed):
        self.random_seed = seed
        self.np_random = get_np_random(seed)

    def generate_seed(self):
        return int(self.np_random.randint(0, self.MAX_RAND_INT))
