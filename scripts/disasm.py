"""Disassembly of one gfx950 kernel out of a HIP object / shared library (llvm-objdump over the offload bundle).
    python scripts/disasm.py kaolin-wisp_amd/csrc/nerf_mlp_bf16.o mlp_bwd2_kernelI14 > /tmp/k.s"""
import os, re, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import kernel_meta as km

path, key = sys.argv[1], sys.argv[2]
objdump = os.path.join(os.path.dirname(km.READELF), "llvm-objdump")
sections = subprocess.run([km.READELF, "-S", "-W", path], capture_output=True, text=True, check=True).stdout
row = next(line.split() for line in sections.splitlines() if ".hip_fatbin" in line)
at = row.index(".hip_fatbin")
offset, size = int(row[at + 3], 16), int(row[at + 4], 16)
blob = open(path, "rb").read()[offset:offset + size]
pos = blob.find(km.MAGIC)
while pos >= 0:
    count = struct.unpack_from("<Q", blob, pos + len(km.MAGIC))[0]
    cursor = pos + len(km.MAGIC) + 8
    for _ in range(count):
        eo, es, tl = struct.unpack_from("<QQQ", blob, cursor)
        cursor += 24
        triple = blob[cursor:cursor + tl].decode()
        cursor += tl
        if "gfx950" not in triple or es == 0:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as tmp:
            tmp.write(blob[pos + eo:pos + eo + es])
        text = subprocess.run([objdump, "-d", "--no-show-raw-insn", tmp.name], capture_output=True, text=True).stdout
        os.unlink(tmp.name)
        on = False
        for line in text.splitlines():
            head = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if head:
                on = key in head.group(1)
            if on:
                print(line)
    pos = blob.find(km.MAGIC, pos + len(km.MAGIC))
