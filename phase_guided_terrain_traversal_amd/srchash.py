"""SHA-256 over the sources physics_kernel is built from (stdlib only: csrc/Makefile runs this file to embed the hash in libpgtt.so, native.py
imports it).  What the translation unit csrc/pgtt_physics_inst.hip includes, plus the Makefile that holds its flags, with comments and white space
removed (a comment edit does not change the kernel)."""
import hashlib
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_sha256() -> str:
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = sorted([os.path.join(_HERE, "csrc", f) for f in ("pgtt_physics_inst.hip", "pgtt_physics.hip.h", "pgtt_physics_quad.hip.h", "pgtt_kernels.hip.h", "Makefile")]
                   + [os.path.join(root, "include", "pgtt.h")])
    for f in files:
        with open(f, "r") as fh:
            text = fh.read()
        if f.endswith("Makefile"):
            text = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("#"))
        else:
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
        h.update(os.path.basename(f).encode() + b"\0" + " ".join(text.split()).encode() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(source_sha256())
