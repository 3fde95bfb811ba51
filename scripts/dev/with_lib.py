"""with_lib.py LIB.so pytest-args...: run pytest with the package bound to another build of the library (development A/B only; the
product never takes the library path from the environment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowgnn_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
