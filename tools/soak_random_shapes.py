import sys, traceback
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_random_shapes as t
bad = 0
for name, fn, lo in (("abi", t.test_random_logits_through_the_c_abi, 10000), ("crop", t.test_random_crops_along_t, 20000),
                     ("joint", t.test_random_additive_joints, 30000)):
    for seed in range(lo, lo + 1500):
        try:
            fn(seed)
        except AssertionError as e:
            bad += 1
            print(name, seed, str(e)[:300].replace("\n", " "), flush=True)
    print(name, "done", flush=True)
print("failures", bad)
