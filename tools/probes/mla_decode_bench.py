"""probe: decode step time of a DeepSeek-V2-Lite-shaped model (MLA, 1 dense + 26 MoE layers, 64 experts top-6, 2 shared) vs position;
synthetic weights, FP16 latent caches.  Not a bench line -- it shows where the MLA arm stands."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig

H, I, E, k, L, V = 2048, 1408, 64, 6, 27, 102400
nh, klr, nd, rd, vhd = 16, 512, 128, 64, 128
kv_max = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = KrasisEngine(device=0); eng.configure(ModelConfig(H, I, E, k, L, 2, 1.0))
eng.fill_synthetic(4, seed=11); eng.set_routing_config("softmax", False, k, E, H)
st = CpuDecodeStore(128, True, False); st.set_moe_store(eng)
rng = np.random.default_rng(3); keep = []; seed = [50]
def W(r, c):
    seed[0] += 1; return st.store_weight_synthetic(r, c, 4, seed[0])
def N(n):
    w = (rng.random(n, dtype=np.float32) * 0.2 + 0.9).astype(np.float32); keep.append(w); return st.store_norm_weight(w.ctypes.data, n)
fin, lm = N(H), W(V, H)
st.configure_decode(H, L, 1e-6, fin, lm, V, k, 1, False, 1.0, 0, synth_seed=5)
half = rd // 2
ang = np.arange(kv_max)[:, None] * (1.0 / 10000.0 ** (2 * np.arange(half) / rd))[None, :]
cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32); keep += [cos, sin]
for l in range(L):
    n_in, n_post = N(H), N(H)
    kv_a, o, q = W(klr + rd, H), W(H, nh * vhd), W(nh * (nd + rd), H)
    w_kc = ((rng.standard_normal((nh, nd, klr)) * 0.06).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    w_vc = ((rng.standard_normal((nh, vhd, klr)) * 0.06).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    kvn = (rng.random(klr) + 0.5).astype(np.float32); keep += [w_kc, w_vc, kvn]
    st.add_decode_mla_layer(n_in, n_post, kv_a, o, q, None, None, w_kc.ctypes.data, w_kc.size, w_vc.ctypes.data, w_vc.size, kvn.ctypes.data, klr, 0, 0,
                            cos.ctypes.data, sin.ctypes.data, half, kv_max, nh, klr, nd, rd, vhd, float(1.0 / np.sqrt(nd + rd)))
    if l == 0:
        st.set_decode_layer_dense(l, W(10944 // 128 * 128, H), W(10944 // 128 * 128, H), W(H, 10944 // 128 * 128))
    else:
        gate = ((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32); gate = (gate.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        eng.set_route_weight_f32(l, gate)
        st.set_decode_layer_moe(l, l, l, W(2 * 2 * I, H), W(H, 2 * I), None)
st.finalize_decode()
st.fill_state_synthetic(kv_max, seed=9)
st.set_use_graph(True)
for i in range(3): st.decode_step(0, 10 + i)
torch.cuda.synchronize()
for pos in (10, 100, 500, kv_max - 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): st.decode_step(0, pos)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pos %5d  %.3f ms/step  (%.1f tok/s)" % (pos, dt / 20 * 1e3, 20 / dt), flush=True)

# prompt pass over the MLA arm (batched projections, per-token attention launches with a token dimension)
P = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if P:
    st.fill_state_synthetic(kv_max, seed=9)
    toks = [int(x) for x in np.random.default_rng(5).integers(0, V, P)]
    st.prefill(toks, 0); torch.cuda.synchronize()
    t0 = time.perf_counter(); st.prefill(toks, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("prompt pass %d tokens: %.1f ms  (%.0f tok/s)" % (P, dt * 1e3, P / dt), flush=True)
