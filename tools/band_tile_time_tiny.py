"""tools/band_tile_time.py with the matrix the reference's evaluation really decodes with: log(p + tiny) (a constant outside the
band; band::band_tile_kernel<2, 12, true>).  python tools/band_tile_time_tiny.py [N] [T]"""
import os, runpy, sys
os.environ['BAND_TILE_TINY'] = '1'
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'band_tile_time.py'), run_name='__main__')
