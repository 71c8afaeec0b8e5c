import sys, os
sys.argv = [sys.argv[0], "316", "3"]
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.zeros(4, device="cuda"); torch.cuda.synchronize()
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lattice_big.py")).read())
