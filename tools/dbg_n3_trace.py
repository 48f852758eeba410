import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tools.bench_next_rows as B
print(B.row_n3())
print(B.row_n1())
