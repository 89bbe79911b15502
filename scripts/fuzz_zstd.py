"""Random frames through wsx_zstd_decode against the bytes they were made from: inputs of many kinds and sizes (skewed bytes of
varying skew -- Huffman literals with one and four streams, streams that end within a period of the bare loop's limits --, text with
matches, runs, random bytes, StreamVByte blocks of random-walk signals, mixtures), compressed by libzstd at random levels, 256 frames
a launch with a guard around every frame's place; then the same frames damaged (a cut, a flipped byte, a wrong declared size): the
decoder may flag them or decode them to something else, it must not write outside a frame's place or hang.
Usage (GPU box): fuzz_zstd.py [launches] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests.test_gpu_zstd import decode_on_device
from tests.test_zstd_oracle import compress

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def one_input():
    kind = rng.integers(0, 8)
    n = int(rng.choice([rng.integers(1, 64), rng.integers(64, 700), rng.integers(700, 9000), rng.integers(9000, 140000), rng.integers(120000, 420000)]))
    if kind == 0:    # skewed bytes: Huffman literals, no matches; the skew decides the code lengths (long codes in a row among them)
        return np.minimum(rng.geometric(rng.uniform(0.02, 0.6), size=n), 255).astype(np.uint8).tobytes()
    if kind == 1:    # a tiny alphabet
        return rng.integers(0, rng.integers(2, 6), size=n).astype(np.uint8).tobytes()
    if kind == 2:    # text with matches
        words = [bytes(rng.integers(97, 123, size=rng.integers(2, 9)).astype(np.uint8)) for _ in range(40)]
        out = b' '.join(words[i] for i in rng.integers(0, 40, size=n // 5 + 1))
        return out[:n]
    if kind == 3:    # runs
        return b''.join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400)) for _ in range(n // 100 + 1))[:n]
    if kind == 4:    # random bytes: raw blocks
        return rng.integers(0, 256, size=n).astype(np.uint8).tobytes()
    if kind == 5:    # what a VBZ chunk holds
        from oracle import vbz
        sig = np.cumsum(rng.integers(-40, 41, size=max(n // 2, 4))).astype(np.int16)
        return vbz.svb_encode(vbz.values_from_samples(sig, True)).tobytes()
    if kind == 6:    # two frequent bytes and a row of rare ones (the longest codes one after the other)
        body = rng.integers(0, 2, size=n).astype(np.uint8)
        if n > 200:
            at = int(rng.integers(0, n - 60))
            body[at:at + 40] = np.arange(60, 100, dtype=np.uint8)
        return body.tobytes()
    a, b = one_input(), one_input()   # a mixture
    return (a + b)[:420000]


t0, n_frames, n_bytes, flagged = time.time(), 0, 0, 0
for launch in range(launches):
    data = [d for d in (one_input() for _ in range(256)) if d]
    frames = [(compress(d, int(rng.choice([-5, 1, 1, 3, 3, 6, 9, 12, 19]))), len(d)) for d in data]
    got, status = decode_on_device(frames)
    for i, (d, g, st) in enumerate(zip(data, got, status)):
        assert st == 0 and g == d, f'launch {launch} frame {i}: status {st}, {len(d)} bytes'
    n_frames, n_bytes = n_frames + len(frames), n_bytes + sum(len(d) for d in data)
    # the same frames, damaged
    bad = []
    for fr, n in frames:
        how = rng.integers(0, 3)
        if how == 0 and len(fr) > 12:
            bad.append((fr[:int(rng.integers(6, len(fr)))], n))
        elif how == 1 and len(fr) > 12:
            b = bytearray(fr)
            b[int(rng.integers(5, len(b)))] ^= int(rng.integers(1, 256))
            bad.append((bytes(b), n))
        else:
            bad.append((fr, max(n + int(rng.integers(-3, 4)), 0)))
    got, status = decode_on_device(bad)
    flagged += int((status != 0).sum())
    print(f'launch {launch}: {len(frames)} frames equal; damaged: {int((status != 0).sum())} flagged, {int((status == 0).sum())} decoded to something', flush=True)
print(f'{n_frames} frames, {n_bytes / 1e6:.1f} MB: all equal to their inputs; {flagged} of {n_frames} damaged frames flagged, nothing written outside a frame\'s place; {time.time() - t0:.0f} s')
