"""Development aid: the pair kernel's parity tests against a variant build (`make -C liftreg_amd/csrc variant NAME=x VFLAGS=…`):
   python tools/run_variant_tests.py x"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip
_hip.LIB_PATH = os.path.join(_hip.CSRC, f"libliftreg_hip_{sys.argv[1]}.so")
import pytest
sys.exit(pytest.main(["tests/test_gpu_conv01_fused.py", "-x", "-q"]))
