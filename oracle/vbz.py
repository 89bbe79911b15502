"""CPU restatement of the VBZ signal decoder -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's checker):
the product never imports this module.

What it restates: upstream reads a read's raw samples with `f5["Raw/Reads"][...]["Signal"][()]` (src/schemas/fast5.py:50-52) and
relies on the HDF5 filter plugin 32020 (ont-vbz-hdf-plugin, a third-party dependency that is NOT in the upstream tree; the upstream
test file test/test_input/test_run1/fast5s/batch_0.fast5 is written with it, version 0, 2-byte integers, zig-zag on).  The
plugin's published chunk layout: u32 little-endian byte count of the decoded samples, then (compression level != 0) one zstd frame;
inside it a StreamVByte block -- ceil(n/4) key bytes, two bits per value (byte length - 1, the first value in the low bits), then
the values' little-endian bytes back to back -- of the differences of consecutive samples (the first against 0), zig-zag mapped
((d << 1) ^ (d >> 15)) when the flag is set; the samples are the running sum in the sample type.

What pins it -- and what does not.  h5py and the filter plugin are absent from every environment this repository has seen, so
NO sample of the upstream test file was ever decoded by upstream's own stack: tests/golden/real_aaat.npz was recorded by
tests/golden/generate_golden.py through THIS repository's reader (warpstr_amd.fast5) and upstream's normalisation + caller on top
of it.  The chain reader -> fixture -> oracle -> kernel is therefore circular as far as the decoding itself goes; the pin is END TO
END ONLY:
  * the ten reads decoded this way call to the README's known answer (44, 40): through the unmodified upstream caller when the
    fixture was recorded (tests/test_fast5.py asserts it of the recorded lengths), through the product from the file itself
    (tests/test_gpu_loci.py, bench.py's from_fast5 leg) -- a sample decoded wrongly moves the whole-read percentiles the
    normalisation depends on;
and these independent, structural facts (tests/test_vbz_oracle.py):
  * a block typed in by hand from the published format text (all four byte lengths, sign changes, the int16 wrap) decodes to the
    samples worked out on paper;
  * for every chunk of the upstream file the zstd frame's declared content size equals ceil(n/4) key bytes + the value bytes those
    keys announce, to the byte; the chunk's u32 header equals 2 x the dataset's length; the sequencer's `duration` attribute of the
    read equals the number of samples; the decoded samples stay inside the DAC range of the device (0..2047)."""
import numpy as np


def svb_block_lengths(keys: np.ndarray, n: int) -> np.ndarray:
    """Byte length of each of the n values a key area describes."""
    codes = np.empty((len(keys), 4), np.uint8)
    for j in range(4):
        codes[:, j] = (keys >> (2 * j)) & 3
    return codes.reshape(-1)[:n].astype(np.int64) + 1


def svb_decode(block: np.ndarray, n: int) -> np.ndarray:
    """n uint32 values of a StreamVByte block (uint8 array); ValueError if the block is shorter than its keys say."""
    block = np.asarray(block, np.uint8)
    n_keys = (n + 3) // 4
    if len(block) < n_keys:
        raise ValueError('StreamVByte block shorter than its key area')
    lens = svb_block_lengths(block[:n_keys], n)
    start = n_keys + np.concatenate([[0], np.cumsum(lens)[:-1]]) if n else np.zeros(0, np.int64)
    if n and int(start[-1] + lens[-1]) > len(block):
        raise ValueError('StreamVByte block shorter than its keys say')
    out = np.zeros(n, np.uint64)
    for b in range(4):
        has = lens > b
        out[has] |= block[start[has] + b].astype(np.uint64) << np.uint64(8 * b)
    return out.astype(np.uint32)


def samples_from_values(values: np.ndarray, zigzag: bool) -> np.ndarray:
    """The int16 samples of a chunk from its StreamVByte values: zig-zag back, running sum, wrap as int16 does."""
    v = values.astype(np.int64)
    d = (v >> 1) ^ -(v & 1) if zigzag else v
    return (np.cumsum(d) & 0xFFFF).astype(np.uint16).view(np.int16)


def decode_block(block: np.ndarray, n: int, zigzag: bool) -> np.ndarray:
    return samples_from_values(svb_decode(block, n), zigzag)


def svb_encode(values: np.ndarray) -> np.ndarray:
    """The StreamVByte block of uint32 values (shortest lengths), for the tests' synthetic streams."""
    values = np.asarray(values, np.uint64)
    n = len(values)
    lens = np.ones(n, np.int64)
    for b in (1, 2, 3):
        lens[values >= (1 << (8 * b))] = b + 1
    keys = np.zeros((n + 3) // 4, np.uint8)
    for i in range(4):
        part = (lens[i::4] - 1).astype(np.uint8)
        keys[:len(part)] |= part << (2 * i)
    start = np.concatenate([[0], np.cumsum(lens)[:-1]]) if n else np.zeros(0, np.int64)
    data = np.zeros(int(lens.sum()), np.uint8)
    for b in range(4):
        has = lens > b
        data[start[has] + b] = ((values[has] >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    return np.concatenate([keys, data])


def values_from_samples(samples: np.ndarray, zigzag: bool) -> np.ndarray:
    """The values the plugin stores for int16 samples (differences in the sample type, zig-zag mapped or taken as unsigned)."""
    s = np.asarray(samples, np.int16).astype(np.int64)
    d = np.diff(np.concatenate([[0], s]))
    d = ((d + 32768) % 65536) - 32768                      # the difference as an int16
    if zigzag:
        return (((d << 1) ^ (d >> 15)) & 0xFFFF).astype(np.uint32)
    return (d & 0xFFFFFFFF).astype(np.uint32)              # (sign-extended to 32 bits: four-byte values for negative differences)
