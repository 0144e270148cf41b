import sys
sys.path.insert(0,'/root/repo/scripts'); sys.argv=[sys.argv[0]]
import importlib.util, types
src=open('/root/repo/scripts/bench_extend.py').read().split('run("8B TP1 Sq4096 x4"')[0]
exec(src)
run("8B 64x256", 32, 8, 16, [0]*64, [256]*64)
run("8B 32x512", 32, 8, 16, [0]*32, [512]*32)
run("8B 16x1024", 32, 8, 16, [0]*16, [1024]*16)
run("8B 8x2048", 32, 8, 16, [0]*8, [2048]*8)
run("8B 16x(1024 on 3072 prefix)", 32, 8, 16, [3072]*16, [1024]*16)
run("8B mixed", 32, 8, 16, [0, 1024, 0, 4096, 0, 0, 2048, 0], [100, 300, 2000, 50, 700, 4000, 128, 1500])
import random
rng = random.Random(5)
inp = [rng.choice([30, 100, 300, 700, 1500, 3000]) for _ in range(32)]
pre = [rng.choice([0, 0, 0, 512, 2048, 8192]) for _ in range(32)]
run("8B ragged 32 requests", 32, 8, 16, pre, inp)
