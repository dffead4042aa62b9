import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = bench.kernel_roofline(B, "bf16", iters=100)
print(os.environ.get("IPOKE_NT_TILE", "auto"), r["avg_launch_us"], "us", r["achieved"], "TF/s")
