timeout 900 python -m pytest tests/test_metrics_gpu.py tests/test_fvd_gpu.py -m gpu -x -q -s 2>&1 | grep -v amdgpu | tail -14
python - <<'PY'
import torch, time
from ipoke_amd import metrics
x=torch.rand(480,3,128,128,device='cuda'); y=(x+0.1*torch.randn_like(x)).clamp(0,1)
for _ in range(3): metrics.psnr_ssim(y,x)
torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): metrics.psnr_ssim(y,x)
e1.record(); torch.cuda.synchronize(); print("psnr+ssim of 480 frames 3x128x128:", e0.elapsed_time(e1)/10, "ms")
PY
