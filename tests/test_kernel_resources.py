"""CPU: register / scratch budgets of the hot kernels, read from the built library's code objects (tests/kernel_meta.py).  The measured
numbers of DESIGN.md section 4 rest on these occupancies; a source or toolchain change that breaks one should fail here, at build time,
not as an unexplained slowdown on the GPU box."""
import os

import pytest

import kernel_meta

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kaolin-wisp_amd", "csrc", "libwisp_hip.so")
pytestmark = pytest.mark.skipif(not kernel_meta.available(LIB), reason="libwisp_hip.so not built or llvm-readelf missing")


@pytest.fixture(scope="module")
def kern():
    meta = kernel_meta.kernels(LIB)
    names = kernel_meta.demangled(list(meta))
    return {names[k]: v for k, v in meta.items()}


def _pick(kern, *parts):
    hits = {n: v for n, v in kern.items() if all(p in n for p in parts)}
    assert hits, parts
    return hits


def test_every_code_object_is_read(kern):
    assert len(kern) > 200 and all(v["wg"] in (64, 128, 256, 512, 1024) for v in kern.values())
    assert all(v["agpr"] <= v["vgpr"] <= 512 for v in kern.values())          # vgpr_count is the unified total, accumulators included


@pytest.mark.parametrize("dtype", ["__hip_bfloat16", "__half"])
def test_flagship_step_kernels_keep_their_occupancy(kern, dtype):
    """nerf_hash.yaml, 16-bit tables (the bench's default): 512-thread workgroups are 2 waves per SIMD each, so <= 128 VGPRs keep two
    of them resident per CU (4 waves / SIMD) - what the queue emitter's capped grid and the decoder's PIN variant are sized for."""
    for name, v in _pick(kern, "hashgrid_bwd_emit_q_kernel<" + dtype + ", 3, 512, 2>").items():
        assert v["scratch"] == 0 and v["vgpr"] <= 96 and v["wg"] == 512, (name, v)           # measured: 87
    # its wide form (launches of >= 2^20 samples): ONE 1024-thread workgroup per CU, i.e. the same 4 waves per SIMD
    for name, v in _pick(kern, "hashgrid_bwd_emit_q_kernel<" + dtype + ", 3, 1024, 2>").items():
        assert v["scratch"] == 0 and v["vgpr"] <= 128 and v["wg"] == 1024, (name, v)         # measured: 87
    # and the one-group-per-wave form of launches below 2^20 samples (round 6): 16 waves of one 64-sample group each
    for name, v in _pick(kern, "hashgrid_bwd_emit_q_kernel<" + dtype + ", 3, 1024, 1>").items():
        assert v["scratch"] == 0 and v["vgpr"] <= 128 and v["wg"] == 1024, (name, v)
    # forward decoder, hidden 64: <TIO, NARROW=false, PIN=true, CODED=false|true>
    for coded in ("false", "true"):
        for name, v in _pick(kern, "mlp_fwd_kernel<" + dtype + ", false, true, " + coded + ">").items():
            assert v["scratch"] == 0 and v["vgpr"] <= 128 and v["wg"] == 512, (name, v)       # measured: 118 / 122
    for name, v in _pick(kern, "mlp_bwd_kernel<" + dtype + ", false,").items():
        assert v["scratch"] == 0 and v["vgpr"] <= 256, (name, v)                               # one 512-thread workgroup per CU
    for name, v in _pick(kern, "hashgrid_fwd_kernel<" + dtype + ", 16, 3>").items():
        assert v["scratch"] == 0 and v["vgpr"] <= 128 and v["wg"] == 256, (name, v)           # measured: 124
    for name, v in _pick(kern, "hashgrid_bwd_reduce_kernel<" + dtype + ", 2, false>").items():
        assert v["scratch"] == 0 and v["vgpr"] <= 64 and v["wg"] == 1024, (name, v)           # measured: 40
    for name, v in _pick(kern, "hashgrid_bwd_reduce_kernel<" + dtype + ", 2, true>").items():     # with the AdamW step in its flush
        assert v["scratch"] == 0 and v["vgpr"] <= 96 and v["wg"] == 1024, (name, v)           # measured: 68


def test_spills_stay_off_the_measured_paths(kern):
    """Some instantiations do spill (feature width 16 hash tables, the decoder without PIN, the pre-queue emitter); none of them is
    launched by the configurations of BASELINE.json.  The list may shrink, not grow."""
    spilling = sorted(n for n, v in kern.items() if v["scratch"] > 0)
    allowed = ("hashgrid_bwd_emit_kernel<", "hashgrid_bwd_kernel<", "hashgrid_bwd_reduce_kernel<float, 16, false>", "hashgrid_bwd_reduce_kernel<__half, 16, false>",
               "hashgrid_bwd_reduce_kernel<__hip_bfloat16, 16, false>", "mlp_fwd_kernel<float, false, false", "mlp_fwd_kernel<__half, false, false",
               "mlp_fwd_kernel<__hip_bfloat16, false, false", "mlp_fwd_kernel<float, true, false", "mlp_fwd_kernel<__half, true, false",
               "mlp_fwd_kernel<__hip_bfloat16, true, false")
    for name in spilling:
        assert any(a in name for a in allowed), name
    assert len(spilling) <= 15 and max(kern[n]["scratch"] for n in spilling) <= 1024
    hot = ("raymarch", "composite", "adamw", "optim", "spc_", "sdf_trace", "codebook", "trilinear", "gather_rows", "nerf_mlp_reduce", "wide")
    for name, v in kern.items():
        if any(h in name for h in hot):
            assert v["scratch"] == 0, (name, v)


def test_hot_kernels_use_the_cdna4_instructions_the_design_names():
    """ISA of the built library (llvm-objdump, no GPU): the decoders run on the matrix cores with gfx950's 16-deep bf16 MFMA and feed
    them with 128-bit LDS reads; the queue emitter's segmented scan is DPP fused multiply-adds with packed fp32 products; the hash-grid
    forward blends with packed FMAs; the 16-bit corner-query backward uses packed atomics.  Presence and order of magnitude only."""
    isa = kernel_meta.instruction_counts(LIB, ("mlp_fwd_kernelI14__hip_bfloat16Lb0ELb1E", "mlp_bwd_kernelI14__hip_bfloat16Lb0E",
                                               "wide_fwd_kernelILi128E14__hip_bfloat16", "wide_chain_kernelILi128E14__hip_bfloat16",
                                               "wide_dw_kernelILi128E14__hip_bfloat16", "wide_dw2_kernelILi128E14__hip_bfloat16",
                                               "hashgrid_bwd_emit_q_kernelI14__hip_bfloat16Li3E",
                                               "hashgrid_fwd_kernelI14__hip_bfloat16Li16ELi3E", "hashgrid_query_kernelI14__hip_bfloat16Lb1E",
                                               "hashgrid_query_kernelI6__halfLb1E"))

    def of(part):
        hits = [c for n, c in isa.items() if part in n]
        assert hits, part
        return hits

    def mfma(c):
        return sum(v for k, v in c.items() if k.startswith("v_mfma_f32") and k.endswith("_bf16"))

    for c in of("mlp_fwd_kernel"):
        assert c["v_mfma_f32_32x32x16_bf16"] >= 20 and c["ds_read_b128"] >= 16 and not any(k.startswith("scratch_") for k in c)
    for c in of("mlp_bwd_kernel"):
        assert mfma(c) >= 60 and c["v_mfma_f32_16x16x32_bf16"] >= 20 and c["ds_read_b128"] >= 32
    for part, least in (("wide_fwd_kernel", 48), ("wide_chain_kernel", 96), ("wide_dw_kernel", 96), ("wide_dw2_kernel", 150)):
        for c in of(part):
            assert mfma(c) >= least, (part, mfma(c))
    for c in of("wide_dw2_kernel"):
        # the pipelined dW kernel (round 5): the producers' 32x32x16 chain (prologue + the pieces under the five stages) and the
        # consumers' 120 16x16x32 products of a round, fed by transposing LDS reads; ten barriers per round + the prologue's
        assert c["v_mfma_f32_16x16x32_bf16"] >= 120 and c["v_mfma_f32_32x32x16_bf16"] >= 100 and c["ds_read_b64_tr_b16"] >= 200
        assert 20 <= c["s_barrier"] <= 26 and not any(k.startswith("scratch_") for k in c)
    for c in of("hashgrid_bwd_emit_q_kernel"):
        dpp = sum(v for k, v in c.items() if k.endswith("_dpp"))
        # (~96 DPP steps per 64-sample group of a wave: 192 in the two-group forms, 98 in the one-group form of small launches)
        assert dpp >= 90 and c["v_pk_mul_f32"] >= 4 and not any(k.startswith("scratch_") for k in c)
    assert max(sum(v for k, v in c.items() if k.endswith("_dpp")) for c in of("hashgrid_bwd_emit_q_kernel")) >= 180
    for c in of("hashgrid_fwd_kernel"):
        assert c["v_pk_fma_f32"] >= 32
    assert all(c["global_atomic_pk_add_bf16"] >= 1 for c in of("hashgrid_query_kernelI14__hip_bfloat16"))
    assert all(c["global_atomic_pk_add_f16"] >= 1 for c in of("hashgrid_query_kernelI6__half"))
