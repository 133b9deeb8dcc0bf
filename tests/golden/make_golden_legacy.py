"""Generate tests/golden/golden_legacy_v1.npz: frames whose huff0 tree descriptions are written the way REAL zipnn wheels
write them.

The reference builds huff0 from the FiniteStateEntropy submodule (/root/reference/setup.py:23-28, .gitmodules:4-6 — not
vendored, not pinned).  That library's weight coder (HUF_compressWeights -> FSE_normalizeCount without the later
`useLowProbCount` parameter) writes a weight count that rounds below one FSE cell as -1; zstd >= 1.4.7 — this
repository's encoder pin — writes +1.  Nearly every chunk's tree description holds such weights (any code
length used by <= 3 symbols), so the -1 form is what every file produced by the PyPI wheel carries.  Both forms are
valid huff0 and every huff0 decoder reads both; this fixture pins that for the decoders here.

Producer: the CPU oracle (oracle/zn_oracle.c) with `zo_set_weight_low_prob(-1)`.  Checked at generation time, in this
container: every frame is decoded by oracle/_ref (the reference csrc/ compiled from where it lies + libzstd 1.4.8's huff0
decoder) and gives the input back; every frame differs from the +1 frame of the same input.

    python tests/golden/make_golden_legacy.py        # rewrites golden_legacy_v1.npz
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

KB = 1024
# name -> (generator kind, bytes, planes, bits_mode, bytes_mode, chunk, seed): full chunks + a ragged tail per dtype, the
# product's real chunk sizes for bf16 / fp8 (256 KiB / 128 KiB) and small ones that keep the fixture small
CASES = {
    "bf16_256k_2chunks_tail": ("bf16", 2 * 256 * KB + 31338, 2, 1, 10, 256 * KB, 41),
    "bf16_64k_3chunks_tail": ("bf16", 3 * 64 * KB + 1002, 2, 1, 10, 64 * KB, 42),
    "fp16_64k_3chunks_tail": ("fp16", 3 * 64 * KB + 6, 2, 0, 10, 64 * KB, 43),
    "fp32_64k_3chunks_tail": ("fp32", 3 * 64 * KB + 1000, 4, 1, 220, 64 * KB, 44),
    "fp8_128k_2chunks_tail": ("fp8", 2 * 128 * KB + 12345, 1, 1, 10, 128 * KB, 45),
    "fp8_64k_2chunks": ("fp8", 2 * 64 * KB, 1, 1, 10, 64 * KB, 46),
}


def main():
    import oracle_lib as O
    from test_oracle import gen_bytes
    assert O.ref_core() is not None, "oracle/_ref is needed to validate the fixture (build container only)"
    hdr = bytes(range(32))
    arrays, meta = {}, []
    for name, (kind, nb, P, rot, bm, chunk, seed) in CASES.items():
        d = gen_bytes(kind, nb, seed)
        plain = O.compress_frame(hdr, d, P, rot, bm, chunk)
        with O.legacy_weights():
            frame = O.compress_frame(hdr, d, P, rot, bm, chunk)
        assert frame != plain, name
        assert O.ref_decompress_body(frame[32:], P, rot, bm, chunk, nb, 2) == d, name       # the reference's C core + libzstd huff0
        assert O.decompress_body(frame[32:], P, rot, bm, chunk, nb) == d, name
        arrays[name + ".frame"] = np.frombuffer(frame, dtype=np.uint8)
        meta.append({"name": name, "kind": kind, "in_len": nb, "num_buf": P, "bits_mode": rot, "bytes_mode": bm, "chunk": chunk,
                     "seed": seed, "in_sha256": hashlib.sha256(d).hexdigest(), "frame_sha256": hashlib.sha256(frame).hexdigest(),
                     "plus1_frame_sha256": hashlib.sha256(plain).hexdigest()})
    arrays["meta.json"] = np.frombuffer(json.dumps(meta, indent=1).encode(), dtype=np.uint8)
    out = os.path.join(HERE, "golden_legacy_v1.npz")
    np.savez_compressed(out, **arrays)
    print(out, os.path.getsize(out), "bytes,", len(meta), "frames")


if __name__ == "__main__":
    main()
