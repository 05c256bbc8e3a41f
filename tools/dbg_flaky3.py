import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_backward as tb
import numpy.testing as npt
orig = npt.assert_allclose
def patched(a, b, *args, **kw):
    try:
        orig(a, b, *args, **kw)
        print("   ok  ", kw.get("err_msg"))
    except AssertionError:
        print("   BAD ", kw.get("err_msg"), "maxabs", float(np.abs(a - b).max()))
np.testing.assert_allclose = patched
tb.np.testing.assert_allclose = patched
case = [x for x in tb.LAYER_CASES if x[:3] == (32, 64, 17)][0]
tb.test_layer_backward(*case)
print("---- 8,8,25")
case = [x for x in tb.LAYER_CASES if x[:3] == (8, 8, 25)][0]
tb.test_layer_backward(*case)
