"""Mutation fuzz of the NetCDF-4 / HDF5 reader over the library-written golden files (tests/golden/hdf5): byte and bit flips, 8-byte
fields overwritten with noise or all-ones, truncation.  Every outcome must be an array or a VrmError -- a crash ends the process.

    tools/ubsan_host.sh && VRM_LIB=volume_renderer_amd/variants/libvrm_ubsan.so python tools/hdf5_mutation_fuzz.py [seed] [iterations]

Round 3: seeds 1, 2, 3 (4000 each) and 11, 12 (6000 each) ran clean under the UndefinedBehaviorSanitizer build (traps on UB).
tests/test_hdf5_golden_cpu.py::test_damaged_library_files_are_read_or_refused runs 600 mutations with every CPU suite."""
import glob
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volume_renderer_amd import VrmError, netcdf  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    iterations = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    rng = np.random.default_rng(seed)
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "hdf5", "*.nc")))
    read = refused = 0
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "mut.nc")
        for it in range(iterations):
            with open(files[it % len(files)], "rb") as f:
                b = bytearray(f.read())
            for _ in range(int(rng.integers(1, 6))):
                mode, i = int(rng.integers(0, 3)), int(rng.integers(0, len(b)))
                if mode == 0:
                    b[i] = int(rng.integers(0, 256))
                elif mode == 1:
                    b[i] ^= 1 << int(rng.integers(0, 8))
                else:
                    j = int(rng.integers(0, len(b) - 8))
                    b[j:j + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8)) if rng.random() < 0.5 else b"\xff" * 8
            if rng.random() < 0.1:
                b = b[:int(rng.integers(8, len(b)))]
            with open(p, "wb") as f:
                f.write(bytes(b))
            try:
                netcdf.read_color(p)
                read += 1
            except (VrmError, MemoryError):
                refused += 1
    print(f"hdf5 mutation fuzz, seed {seed}: {read} read, {refused} refused, 0 crashes")


if __name__ == "__main__":
    main()
