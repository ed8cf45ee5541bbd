#!/usr/bin/env python3
"""Perplexity of a REAL checkpoint through the product path, the reference's protocol (perplexity/measure_ppl.py:154-297: sliding windows of
--window tokens every --stride, first window scores everything, later ones the new tokens only), in the exact mode and in the tolerance modes.
Reproduces the pins of the reference's README (README.md:87-95: Qwen3-Coder-Next 7.23 / 12.52, DeepSeek-V2-Lite 6.03 / 9.22) the day the weights are
on the box -- there are none in this image, so nothing here has been run against a real model.

    python tools/probes/ppl_model.py --model-path /models/Qwen3-Coder-Next --tokens wikitext_tokens.npy [--bits 4] [--window 2048] [--stride 1024]
                                     [--max-tokens 100000] [--modes exact,attn_fast,gemm_fast]

--tokens: a .npy / .bin (int32) file of token ids produced by the reference's tokenizer step (measure_ppl.py tokenises with the model's own tokenizer;
the ids are what both harnesses consume)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--tokens", required=True)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--window", type=int, default=2048)
    ap.add_argument("--stride", type=int, default=1024)
    ap.add_argument("--max-tokens", type=int, default=0)
    ap.add_argument("--modes", default="exact,attn_fast,gemm_fast")
    ap.add_argument("--kv-fp8", action="store_true")
    args = ap.parse_args()
    from krasis_amd import evaluate_perplexity
    from krasis_amd.decode_setup import CpuDecoder
    toks = np.load(args.tokens) if args.tokens.endswith(".npy") else np.fromfile(args.tokens, np.int32)
    toks = [int(t) for t in (toks[: args.max_tokens] if args.max_tokens else toks)]
    dec = CpuDecoder(args.model_path, expert_bits=args.bits, decode_bits=args.bits, kv_fp8=args.kv_fp8)
    dec.prepare(max_seq=args.window + 64)
    out = {}
    for mode in args.modes.split(","):
        dec._store.set_attention_mode(mode in ("attn_fast", "gemm_fast"), gemm_fast=(mode == "gemm_fast"))
        r = evaluate_perplexity(dec._store, toks, args.window, args.stride)
        out[mode] = r
        print("%-10s PPL %.4f  (%d tokens scored, %d windows, %.1f s)" % (mode, r["perplexity"], r["num_tokens_scored"], r["num_windows"], r["elapsed_s"]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
