"""Builds the reference's ONLY native component -- the Cython evaluator torchreid/metrics/rank_cylib/rank_cy.pyx -- from
the sources where they lie under /root/reference into oracle/_ref/ (git-ignored AND gpurun-ignored: it stays in the build
container). Test infrastructure: used there to cross-check oracle.eval_market1501 and to capture its outputs on seeded
inputs as data (tests/golden/rank_market1501_cy.npz, make_golden.py F14) -- what the GPU-side tests compare against; never
imported by the product, never loaded on the GPU box. Recipe = what the reference's own setup.py does (cythonize + one C compile), driven directly:

    cython -3 rank_cy.pyx -o oracle/_ref/rank_cy.c ; gcc -O2 -shared -fPIC $(python-config --includes) -I numpy ...

No reference source is copied into the repository: only the built .so stays under oracle/_ref/ (the generated C file, which
quotes the .pyx line by line, is deleted after the compile), and neither travels to the GPU box.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PYX = "/root/reference/torchreid/metrics/rank_cylib/rank_cy.pyx"
OUT = os.path.join(HERE, "_ref")


def build(verbose=True):
    """-> path of the built extension, or None when the reference tree / Cython is not available here."""
    if not os.path.exists(REF_PYX):
        return None
    try:
        import Cython  # noqa: F401
        import numpy as np
    except ImportError:
        return None
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "rank_cy" + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(REF_PYX):
        return so
    c_file = os.path.join(OUT, "rank_cy.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", REF_PYX, "-o", c_file])
    cmd = ["gcc", "-O2", "-shared", "-fPIC", "-w", "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
           "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION", c_file, "-o", so]
    subprocess.check_call(cmd)
    os.remove(c_file)   # the intermediate carries the reference's source text as comments: only the binary is kept
    if verbose:
        print("built", so)
    return so


def load():
    """Import oracle/_ref/rank_cy if it has been built (build container only); None otherwise."""
    import importlib.util
    import glob
    hits = glob.glob(os.path.join(OUT, "rank_cy*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("rank_cy", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build())
