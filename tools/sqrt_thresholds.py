#!/usr/bin/env python3
"""Derives the two constants of cgrt_bezier.hpp (newton_unconverged / newton_accept): with a correctly rounded square root,
sqrt(s) > 1e-6 <=> s > A and sqrt(s) < 1e-4 <=> s < B, A = the largest double whose root is <= 1e-6, B = the smallest whose root
is >= 1e-4.  Checks both equivalences on the 2 x 2001 doubles around each edge and on random values (math.sqrt is IEEE)."""
import math
import random


def largest_with_root_le(t):
    x = t * t
    while math.sqrt(x) <= t:
        x = math.nextafter(x, math.inf)
    while math.sqrt(x) > t:
        x = math.nextafter(x, -math.inf)
    return x


def smallest_with_root_ge(t):
    x = t * t
    while math.sqrt(x) >= t:
        x = math.nextafter(x, -math.inf)
    while math.sqrt(x) < t:
        x = math.nextafter(x, math.inf)
    return x


A, B = largest_with_root_le(1e-6), smallest_with_root_ge(1e-4)
print("A =", A.hex(), " B =", B.hex())
assert A.hex() == "0x1.19799812dea11p-40" and B.hex() == "0x1.5798ee2308c3ap-27"
for edge, f, g in ((A, lambda s: math.sqrt(s) > 1e-6, lambda s: s > A), (B, lambda s: math.sqrt(s) < 1e-4, lambda s: s < B)):
    x = edge
    for _ in range(1000):
        x = math.nextafter(x, -math.inf)
    for _ in range(2001):
        assert f(x) == g(x), x.hex()
        x = math.nextafter(x, math.inf)
rng = random.Random(1)
for _ in range(200000):
    s = 10.0 ** rng.uniform(-20, 2)
    assert (math.sqrt(s) > 1e-6) == (s > A) and (math.sqrt(s) < 1e-4) == (s < B)
for s in (0.0, math.inf, math.nan):
    assert (math.sqrt(s) > 1e-6) == (s > A) and (math.sqrt(s) < 1e-4) == (s < B)
print("ok")
