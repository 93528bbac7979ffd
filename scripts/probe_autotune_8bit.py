import sys, json, torch
sys.path.insert(0, '/root/repo')
import gemlite_amd
H = gemlite_amd.helper
dev = torch.device("cuda:0")
N = K = 4096
W = (torch.randn(N, K, device=dev) / 30).half()
for name, lin in (("A8W8_int8_dynamic", H.A8W8_int8_dynamic(device=dev, dtype=torch.float16).from_weights(W)),
                  ("A16W8_INT8", H.A16W8(device=dev, dtype=torch.float16).from_weights(W))):
    res = H.autotune_layer(lin, batch_sizes=(8, 64, 256), iters=20, cold=True)
    print(name, json.dumps({M: dict(tuning=v["tuning"], us=v["us"], default_us=v["default_us"]) for M, v in res.items()}))
    x = (torch.randn(64, K, device=dev) / 10).half()
    y = lin(x); torch.cuda.synchronize(); print("ok", float(y.float().abs().mean()))
