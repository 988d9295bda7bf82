import torch, subprocess, sys, os, time
x = torch.empty(int(os.environ.get("HOLD_GB","30")) << 30, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for i in range(2):
    r = subprocess.run([sys.executable, "scripts/npp_train_step_timing.py"], capture_output=True, text=True)
    print("held", os.environ.get("HOLD_GB","30"), r.stdout.strip().splitlines()[-1][:60], flush=True)
