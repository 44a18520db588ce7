import sys, traceback
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_random_shapes as t
# usage: soak_random_shapes.py [cases per surface] [first seed] [logit standard deviations ...]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
if len(sys.argv) > 3:
    t.SCALES = tuple(float(a) for a in sys.argv[3:])
only = os.environ.get("SOAK_ONLY", "").split(",") if os.environ.get("SOAK_ONLY") else None  # e.g. SOAK_ONLY=dense,decode
if os.environ.get("SOAK_BIG"):  # lattices up to 600 frames x 1024 label positions, ten times the cells
    t.MAX_T, t.MAX_U1, t.SIZE_FACTOR = 600, 1024, 10
if os.environ.get("SOAK_VMAX"):  # vocabularies up to this size (default 1200), e.g. 4500 for the widest row kernels
    d = list(t._draw.__defaults__)
    d[2] = int(os.environ["SOAK_VMAX"])
    t._draw.__defaults__ = tuple(d)
bad = 0
for name, fn, lo in (("abi", t.test_random_logits_through_the_c_abi, first), ("crop", t.test_random_crops_along_t, first + 100000),
                     ("joint", t.test_random_additive_joints, first + 200000), ("dense", t.test_random_dense_layers, first + 300000),
                     ("fc_joint", t.test_random_fc_joint_heads, first + 400000), ("decode", t.test_random_decoder_steps, first + 500000),
                     ("malformed", t.test_random_malformed_lengths_and_labels, first + 600000),
                     ("malformed_joint", t.test_random_malformed_inputs_to_the_fused_joint, first + 700000)):
    if only and name not in only:
        continue
    for seed in range(lo, lo + n):
        try:
            fn(seed)
        except AssertionError as e:
            bad += 1
            print(name, seed, str(e)[:300].replace("\n", " "), flush=True)
        if (seed - lo) % 25 == 24:
            print(name, seed - lo + 1, "cases", flush=True)  # a long run shows that it is alive
    print(name, "done", flush=True)
print("failures", bad)
