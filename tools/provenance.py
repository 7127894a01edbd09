"""What the replayed counter files were measured on: a hash of the kernel sources and of the built library.

`profiles/pmc_latest.json` (written by tools/pmc_summary.py from separate rocprofv3 --pmc passes) is replayed by bench.py
as `roofline.traffic` / `roofline.valu`.  Both sides call these functions; bench.py sets the replayed fields to null, with
the reason, when the kernel sources differ from the ones the counters were collected on.  The SOURCE hash is the one
compared: the library is rebuilt by `build()` wherever the repository is checked out and its bytes are not guaranteed to
be identical across builds; the library hash is recorded as additional information.
"""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha256():
    """sha256 over the files libgsraster.so is built from (names and contents, in sorted order)."""
    csrc = os.path.join(ROOT, "gscream_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")) or f == "Makefile")
    files.append(os.path.join(ROOT, "include", "gsraster.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def library_sha256(path=None):
    path = path or os.environ.get("GSR_LIB") or os.path.join(ROOT, "gscream_amd", "libgsraster.so")
    if not os.path.exists(path):
        return None
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:  # noqa: BLE001  (no .git on the GPU box: the snapshot travels without history)
        return None


def stamp():
    return {"kernel_source_sha256": kernel_source_sha256(), "library_sha256": library_sha256(), "git_head": git_head()}
