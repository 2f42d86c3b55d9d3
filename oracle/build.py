"""Oracle (test infrastructure): build recipe for the C part of the oracle.

``build_c()`` compiles ``oracle/nms_ref.c`` -> ``oracle/_build/liboracle_nms.so`` with gcc.
There is no ``oracle/_ref`` (a build of the reference itself): the reference is pure Python on
TensorFlow -- nothing in it compiles, and TensorFlow is absent (see DESIGN.md "Oracle").
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build_c(force=False):
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(HERE, "nms_ref.c")
    out = os.path.join(out_dir, "liboracle_nms.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_c(force=True))
