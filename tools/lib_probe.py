"""Headline bench line with an alternative build of the library: python tools/lib_probe.py tools/libX.so [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torbi_amd._lib as _lib
_lib.LIBRARY = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
