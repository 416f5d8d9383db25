import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from align3d_amd import Context, R3dTree
from data_util import uniform01
ctx = Context(0)
db = uniform01(10, 1500000).reshape(500000, 3)
for _ in range(5):
    R3dTree.new(ctx, db).free()
