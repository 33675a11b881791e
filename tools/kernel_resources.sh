#!/bin/bash
# VGPRs / spills / LDS / code size of the heavy kernels of a kernel library: tools/kernel_resources.sh castro_amd/libcastro_hydro_amd.so
# (reads the code object's metadata notes; no GPU needed)
LIB=${1:-castro_amd/libcastro_hydro_amd.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$LIB > /dev/null 2>&1
# the device code objects sit in the .hip_fatbin section: unbundle every gfx950 entry
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fatbin $LIB 2>/dev/null
python3 - "$T/fatbin" "$T" <<'PY'
import sys, struct, subprocess, re, os
data = open(sys.argv[1], "rb").read()
out = sys.argv[2]
# concatenated clang offload bundles ("__CLANG_OFFLOAD_BUNDLE__")
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = 0; n = 0
rows = []
while True:
    pos = data.find(magic, pos)
    if pos < 0: break
    nb = struct.unpack_from("<Q", data, pos + 24)[0]
    p = pos + 32
    for _ in range(nb):
        off, size, tl = struct.unpack_from("<QQQ", data, p); p += 24
        triple = data[p:p + tl].decode(); p += tl
        if "gfx950" in triple and size:
            f = os.path.join(out, "co%d.elf" % n); n += 1
            open(f, "wb").write(data[pos + off: pos + off + size])
            txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+).*?(?=\n\s+- \.agpr_count|\Z)", txt, re.S):
                pass
            cur = {}
            for line in txt.splitlines():
                s = line.strip()
                if s.startswith("- .agpr_count") or s.startswith("- .args"):
                    if cur.get("name"): rows.append(cur)
                    cur = {}
                for key in (".agpr_count", ".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size", ".private_segment_fixed_size"):
                    if s.lstrip("- ").startswith(key + ":"):
                        cur[key] = int(s.split(":")[1])
                if s.lstrip("- ").startswith(".name:"):
                    cur["name"] = s.split(":", 1)[1].strip()
            if cur.get("name"): rows.append(cur)
    pos += len(magic)
import shutil
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("%-64s %5s %5s %6s %6s %7s" % ("kernel", "vgpr", "agpr", "vspill", "lds", "scratch"))
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void cad::", "").replace("cad::", "")
    if not any(k in d for k in ("k_trace_pair", "k_trace<", "k_riemann1<0", "k_trans1_fold_lds", "k_trans1_tile", "k_final_tile", "k_fab_ops", "k_final<", "k_finalx_consup", "k_ctoprim", "k_divu_pair", "k_hydro")): continue
    print("%-64s %5d %5d %6d %6d %7d" % (d[:64], r.get(".vgpr_count", -1), r.get(".agpr_count", 0), r.get(".vgpr_spill_count", 0), r.get(".group_segment_fixed_size", 0), r.get(".private_segment_fixed_size", 0)))
PY
rm -rf $T
