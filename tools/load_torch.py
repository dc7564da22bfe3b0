import torch, time, sys
a = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20):
        c = a @ b
    torch.cuda.synchronize()
