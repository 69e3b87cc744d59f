#!/usr/bin/env python3
"""Static instruction mix of the hot kernels in libwisp_hip.so (no GPU needed): per kernel the number of MFMA, other vector ALU,
scalar, LDS, global / buffer memory, wait and branch instructions in the disassembly, plus registers / scratch from the metadata.
Static counts, not dynamic ones: loops are counted once.  Usage: python scripts/isa_mix.py [substring ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_meta  # noqa: E402

LIB = os.path.join(ROOT, "kaolin-wisp_amd", "csrc", "libwisp_hip.so")
DEFAULT = ("hashgrid_fwd_kernelI14__hip_bfloat16Li16ELi3E", "hashgrid_bwd_emit_q_kernelI14__hip_bfloat16Li3E",
           "hashgrid_bwd_reduce_kernelI14__hip_bfloat16Li2E", "mlp_fwd_kernelI14__hip_bfloat16Lb0ELb1E", "mlp_bwd_kernelI14__hip_bfloat16Lb0E",
           "wide_fwd_kernelILi128E14__hip_bfloat16", "wide_chain_kernelILi128E14__hip_bfloat16", "wide_dw_kernelILi128E14__hip_bfloat16",
           "raymarch_ray", "composite", "adamw_groups", "sdf_trace_fused_kernelI14__hip_bfloat16", "spc_trilinear_multi_fwd")


def bucket(mnemonic):
    if mnemonic.startswith("v_mfma") or mnemonic.startswith("v_smfma"):
        return "mfma"
    if mnemonic.startswith("s_waitcnt") or mnemonic.startswith("s_wait"):
        return "wait"
    if mnemonic.startswith("s_cbranch") or mnemonic.startswith("s_branch"):
        return "branch"
    if mnemonic.startswith("ds_"):
        return "lds"
    if mnemonic.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "mem"
    if mnemonic.startswith("v_"):
        return "valu"
    if mnemonic.startswith("s_"):
        return "salu"
    return "other"


def main():
    wanted = tuple(sys.argv[1:]) or DEFAULT
    meta = kernel_meta.kernels(LIB)
    isa = kernel_meta.instruction_counts(LIB, wanted)
    pretty = kernel_meta.demangled(list(isa))
    print(f"{'total':>6} {'mfma':>5} {'valu':>6} {'salu':>5} {'lds':>5} {'mem':>5} {'wait':>5} {'br':>4} {'vgpr':>5} {'scr':>4}  kernel")
    for name in sorted(isa, key=lambda n: pretty[n]):
        c = isa[name]
        b = {}
        for mnemonic, n in c.items():
            b[bucket(mnemonic)] = b.get(bucket(mnemonic), 0) + n
        m = meta.get(name, {})
        print(f"{sum(c.values()):6d} {b.get('mfma', 0):5d} {b.get('valu', 0):6d} {b.get('salu', 0):5d} {b.get('lds', 0):5d} {b.get('mem', 0):5d} "
              f"{b.get('wait', 0):5d} {b.get('branch', 0):4d} {m.get('vgpr', -1):5d} {m.get('scratch', -1):4d}  {pretty[name][:120]}")


if __name__ == "__main__":
    main()
