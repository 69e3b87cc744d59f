"""Test infrastructure: per-kernel resource metadata (VGPRs, SGPRs, scratch, LDS, workgroup size) of the gfx950 code objects inside
libwisp_hip.so, read without a GPU: the `.hip_fatbin` section is a sequence of clang offload bundles (one per translation unit); each
gfx950 entry is an ELF whose AMDGPU notes list `amdhsa.kernels`."""
import os
import re
import struct
import subprocess
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def available(lib_path):
    return os.path.isfile(lib_path) and os.path.isfile(READELF)


def kernels(lib_path):
    """{mangled kernel name: dict(vgpr, agpr, sgpr, scratch, lds, wg)}"""
    sections = subprocess.run([READELF, "-S", "-W", lib_path], capture_output=True, text=True, check=True).stdout
    row = next(line.split() for line in sections.splitlines() if ".hip_fatbin" in line)
    at = row.index(".hip_fatbin")
    offset, size = int(row[at + 3], 16), int(row[at + 4], 16)
    with open(lib_path, "rb") as f:
        f.seek(offset)
        blob = f.read(size)
    found = {}
    pos = blob.find(MAGIC)
    while pos >= 0:
        count = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        cursor = pos + len(MAGIC) + 8
        for _ in range(count):
            entry_offset, entry_size, triple_len = struct.unpack_from("<QQQ", blob, cursor)
            cursor += 24
            triple = blob[cursor:cursor + triple_len].decode()
            cursor += triple_len
            if "gfx950" not in triple or entry_size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as tmp:
                tmp.write(blob[pos + entry_offset:pos + entry_offset + entry_size])
            try:
                notes = subprocess.run([READELF, "--notes", tmp.name], capture_output=True, text=True, check=True).stdout
            finally:
                os.unlink(tmp.name)
            for block in notes.split("- .agpr_count:")[1:]:
                field = lambda key: re.search(r"\." + key + r":\s+(\S+)", block).group(1)      # noqa: E731
                found[field("name")] = dict(vgpr=int(field("vgpr_count")), agpr=int(block.split()[0]), sgpr=int(field("sgpr_count")),
                                            scratch=int(field("private_segment_fixed_size")), lds=int(field("group_segment_fixed_size")),
                                            wg=int(field("max_flat_workgroup_size")))
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return found


def demangled(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def instruction_counts(lib_path, kernel_substrings):
    """{mangled kernel name: Counter(mnemonic)} for the kernels whose mangled name contains one of `kernel_substrings`
    (llvm-objdump -d of each gfx950 code object)."""
    import collections
    objdump = os.path.join(os.path.dirname(READELF), "llvm-objdump")
    sections = subprocess.run([READELF, "-S", "-W", lib_path], capture_output=True, text=True, check=True).stdout
    row = next(line.split() for line in sections.splitlines() if ".hip_fatbin" in line)
    at = row.index(".hip_fatbin")
    offset, size = int(row[at + 3], 16), int(row[at + 4], 16)
    with open(lib_path, "rb") as f:
        f.seek(offset)
        blob = f.read(size)
    counts = {}
    pos = blob.find(MAGIC)
    while pos >= 0:
        count = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        cursor = pos + len(MAGIC) + 8
        for _ in range(count):
            entry_offset, entry_size, triple_len = struct.unpack_from("<QQQ", blob, cursor)
            cursor += 24
            triple = blob[cursor:cursor + triple_len].decode()
            cursor += triple_len
            if "gfx950" not in triple or entry_size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as tmp:
                tmp.write(blob[pos + entry_offset:pos + entry_offset + entry_size])
            try:
                text = subprocess.run([objdump, "-d", "--no-show-raw-insn", tmp.name], capture_output=True, text=True, check=True).stdout
            finally:
                os.unlink(tmp.name)
            current = None
            for line in text.splitlines():
                head = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if head:
                    current = head.group(1) if any(k in head.group(1) for k in kernel_substrings) else None
                    continue
                if current:
                    ins = re.match(r"^\s+([a-z][a-z0-9_]+)", line)
                    if ins:
                        counts.setdefault(current, collections.Counter())[ins.group(1)] += 1
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return counts
