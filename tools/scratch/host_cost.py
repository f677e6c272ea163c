import sys, time, torch, cProfile, pstats
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device("cuda", 0)
model, _ = bench.build_model("bf16", dev)
x = torch.randn(4, 120, 80, device=dev).to(torch.bfloat16)
lens = torch.tensor([120, 100, 90, 64], dtype=torch.int32, device=dev)
def run(n):
    with torch.no_grad():
        for _ in range(n):
            enc, mask = model._forward_encoder(x, lens)
            lp = model.ctc_logprobs(enc)
run(5); torch.cuda.synchronize()
t0 = time.perf_counter(); run(50); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time per pass {1e3*(t1-t0)/50:.3f} ms, incl. drain {1e3*(t2-t0)/50:.3f} ms")
pr = cProfile.Profile(); pr.enable(); run(30); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
