"""Content hash of the sources place_batch_kernel is compiled from (the kernel header and what it includes).  A PMC
summary under profiles/ is stamped with it (tools/pmc_summary.py); bench.py takes a summary's traffic figure only when the
stamp equals the hash of the tree it runs on, and says so otherwise (`roofline.traffic_provenance`)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("place_kernel.hpp", "snapshot.hpp", "wave.hpp")


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in FILES:
        h.update(open(os.path.join(ROOT, "modelmesh_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
