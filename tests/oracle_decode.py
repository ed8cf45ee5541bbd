"""Oracle-side decode_step driver: the control flow of src/decode.rs:2690-3520 over the oracle's operator restatements.
Pure test infrastructure (slow, small models only)."""
import numpy as np

from oracle import oracle as O

F = np.float32


class OracleDecode:
    def __init__(self, hidden, eps, norm_bias_one, topk, scoring, norm_topk, rsf, embedding, vocab):
        self.H, self.eps, self.nbo = hidden, F(eps), norm_bias_one
        self.topk, self.scoring, self.norm_topk, self.rsf = topk, scoring, norm_topk, F(rsf)
        self.emb, self.vocab = embedding, vocab
        self.weights, self.norms, self.layers = [], [], []
        self.rope = None
        self.hidden = np.zeros(hidden, F); self.residual = np.zeros(hidden, F)

    def store_weight_f32(self, w, bits=4):
        rows, cols = w.shape
        pt, st = (O.quantize_f32_to_transposed_int4(w) if bits == 4 else O.quantize_f32_to_transposed_int8(w))
        self.weights.append((pt, st, rows, cols, bits)); return len(self.weights) - 1

    def store_weight_packed(self, pt, st, rows, cols, bits=4):
        self.weights.append((pt, st, rows, cols, bits)); return len(self.weights) - 1

    def store_norm(self, w):
        self.norms.append(np.ascontiguousarray(w, F)); return len(self.norms) - 1

    def matvec(self, wid, x):
        pt, st, rows, cols, bits = self.weights[wid]
        q, s = O.quant_act_int16_f32(np.ascontiguousarray(x[:cols], F))
        return O.matvec_int4_t(pt, st, q, s) if bits == 4 else O.matvec_int8_t(pt, st, q, s)

    def moe_block(self, L, hidden, ids_w=None):
        """decode.rs:3286-3402: router on the f32 hidden; routed experts see bf16(hidden) (:3307-3309) and are summed in routing order by
        moe_forward_unified; `moe_output *= rsf` when rsf != 1 (:3338-3340) -- the ROUTED sum only; the shared expert reads the F32 hidden
        (:3356-3359), fast_silu_mul + f32::round digits (:3364-3374), and is scaled by 1 / (1 + exp(-gate)) with libm exp (:3379-3393);
        hidden = moe_output + shared_out (:3396-3399).  ids_w: (ids, weights) to bypass the router (known-answer tests)."""
        if ids_w is None:
            ids, w, _ = O.route_decode(L["gate"], hidden, self.topk, self.scoring, self.norm_topk, L.get("bias"), L.get("esc"))
        else:
            ids, w = ids_w
        act = O.f32_to_bf16(hidden)
        sel = [L["experts"][i] for i in ids]
        if sel and isinstance(sel[0], O.GgufExpert):      # native GGUF layer (moe.rs:990 moe_forward_gguf): only the prompt pass takes these
            moe = O.moe_forward_gguf(sel, np.asarray(w, F), act)
        else:
            moe = O.moe_forward_unified(sel, np.asarray(w, F), act)
        if self.rsf != F(1.0):
            moe = (moe * self.rsf).astype(F)
        if L.get("sgu") is not None:
            gu = self.matvec(L["sgu"], hidden); si = gu.size // 2
            hid = O.fast_silu_mul(gu[:si], gu[si:])
            sh = self.matvec(L["sd"], hid)
            if L.get("sg") is not None:
                gv = self.matvec(L["sg"], hidden)[0]
                sh = (sh * F(O.sigmoid(float(gv), O.SIG_LIBM))).astype(F)
            return (moe + sh).astype(F)
        return moe

    def step(self, token, pos):
        H = self.H
        self.hidden = self.emb[token].copy()
        first = True
        for L in self.layers:
            self.hidden, self.residual = O.fused_add_rmsnorm(self.hidden, self.residual, self.norms[L["in_norm"]], self.eps, first, self.nbo)
            first = False
            if L["attn"] == "la":
                qkvz = self.matvec(L["qkvz"], self.hidden); ba = self.matvec(L["ba"], self.hidden)
                c = O.la_conv(qkvz, ba, L["conv_state"], L["conv_w"], L["a_log"], L["dt_bias"], L["scale"], L["nk"], L["nv"], L["dk"], L["dv"], 4)
                L["conv_state"] = c["conv_state"]
                ro, L["recur_state"] = O.la_recurrent(L["recur_state"], c["q"], c["k"], c["v"], c["g"], c["beta"], L["nv"], L["dk"], L["dv"])
                gated = O.gated_rmsnorm_silu(ro, c["z"], L["norm_w"], L["nv"], L["dv"], self.eps)
                self.hidden = self.matvec(L["out"], gated)
            elif L["attn"] == "gqa":
                q = self.matvec(L["q"], self.hidden); k = self.matvec(L["k"], self.hidden); v = self.matvec(L["v"], self.hidden)
                ao, L["kv_k"], L["kv_v"] = O.gqa_step(q, k, v, L["q_norm"], L["k_norm"], L["gated"], L["nh"], L["nkv"], L["hd"], self.eps,
                                                     self.rope[0], self.rope[1], L["kv_k"], L["kv_v"], pos, L["sm_scale"])
                self.hidden = self.matvec(L["o"], ao)
            elif L["attn"] == "mla":                                         # decode.rs:2993-3252
                kv_out = self.matvec(L["kv_a"], self.hidden)
                if L.get("q") is not None:
                    q_full = self.matvec(L["q"], self.hidden)
                else:
                    qc = self.matvec(L["q_a"], self.hidden)
                    if L.get("q_a_norm") is not None:
                        qc = O.rmsnorm_seq(qc, L["q_a_norm"], self.eps)
                    pad = np.zeros(self.weights[L["q_b"]][3], F); pad[: qc.size] = qc
                    q_full = self.matvec(L["q_b"], pad)
                vp, L["ckv"], L["kpe"] = O.mla_step(kv_out, q_full, L["kv_a_norm"], L["w_kc"], L["w_vc"], L["cos"], L["sin"], L["nh"], L["klr"],
                                                   L["nd"], L["rd"], L["vhd"], self.eps, L["sm_scale"], L["ckv"], L["kpe"], pos)
                self.hidden = self.matvec(L["o"], vp)
            self.hidden, self.residual = O.fused_add_rmsnorm(self.hidden, self.residual, self.norms[L["post_norm"]], self.eps, False, self.nbo)
            if L.get("mlp") == "moe":
                self.hidden = self.moe_block(L, self.hidden)
            elif L.get("mlp") == "dense":
                g = self.matvec(L["gate_w"], self.hidden); u = self.matvec(L["up_w"], self.hidden)
                K = self.weights[L["down_w"]][3]
                hid = np.zeros(K, F); hid[: g.size] = O.fast_silu_mul(g, u)
                self.hidden = self.matvec(L["down_w"], hid)
        self.hidden, self.residual = O.fused_add_rmsnorm(self.hidden, self.residual, self.norms[self.final_norm], self.eps, False, self.nbo)
        return self.matvec(self.lm_head, self.hidden)
