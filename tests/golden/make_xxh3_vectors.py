"""Generates tests/golden/xxh3_vectors.json with the Python `xxhash` package (libxxhash 0.8.2),
the public XXH3-64 implementation available in the build container.  These pin the oracle's
(and the HIP kernel's) XXH3 restatement: xxhash-rust 0.8.6 `xxh3_64` (src/uniq.rs:45 of the
reference) implements the same published function, seed 0, default secret.
Run:  python tests/golden/make_xxh3_vectors.py
"""
import json, os, random
import xxhash

rng = random.Random(20261003)
lens = list(range(0, 260)) + [511, 512, 513, 1000, 1023, 1024, 1025, 1087, 1088, 1089, 2047, 2048, 2049,
                               3000, 4096, 5000, 20000]
vecs = []
for n in lens:
    s = "".join(rng.choice("ACGT") for _ in range(n))
    vecs.append({"in": s, "xxh3_64": "%016x" % xxhash.xxh3_64_intdigest(s.encode())})
for s in ["", "AAAAAAAT", "A" * 1000, "banana", "N-N-N", "ACGTN-" * 100]:
    vecs.append({"in": s, "xxh3_64": "%016x" % xxhash.xxh3_64_intdigest(s.encode())})
# arbitrary bytes (latin-1 encoded in the JSON)
for n in [1, 2, 3, 4, 7, 8, 9, 16, 17, 129, 241, 777]:
    b = bytes(rng.randrange(256) for _ in range(n))
    vecs.append({"in_latin1": b.decode("latin-1"), "xxh3_64": "%016x" % xxhash.xxh3_64_intdigest(b)})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xxh3_vectors.json")
json.dump({"generator": "xxhash %s / libxxhash %s" % (xxhash.VERSION, xxhash.XXHASH_VERSION), "vectors": vecs},
          open(out, "w"), indent=0)
print("wrote", out, len(vecs))
