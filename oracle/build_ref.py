"""Builds the one piece of the reference's own geometry code that compiles stand-alone here:
/root/reference/python/depthmotionnet/dataset_tools/view_tools_cython.pyx (Cython + numpy only).

The .pyx is read where it lies; generated C and the extension module go to oracle/_ref/ (git-ignored).
Nothing is copied into the repository sources.  Used by tests/golden/make_golden.py to pin the
depth -> flow geometry of the oracle (pixel centre +0.5, X2 = R*X1 + t, projection by K).

TensorFlow 1.4 and lmbspecialops (the rest of the path) are not in /root/reference and cannot be built:
see DESIGN.md "Oracle".
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DEMON_REFERENCE", "/root/reference")
PYX = os.path.join(REF, "python", "depthmotionnet", "dataset_tools", "view_tools_cython.pyx")
OUT_DIR = os.path.join(HERE, "_ref")


def available():
    return os.path.exists(PYX)


def build():
    """Returns the path of oracle/_ref/view_tools_cython*.so, or None when /root/reference is absent."""
    if not available():
        return None
    import numpy

    os.makedirs(OUT_DIR, exist_ok=True)
    c_file = os.path.join(OUT_DIR, "view_tools_cython.c")
    so = os.path.join(OUT_DIR, "view_tools_cython" + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(PYX):
        return so
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "-o", c_file, PYX])
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fwrapv", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                           "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(), c_file, "-o", so])
    return so


if __name__ == "__main__":
    print(build())
