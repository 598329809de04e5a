#!/usr/bin/env python3
"""One-screen summary of a kernel's instruction stream, from the SHIPPED library's disassembly (no GPU needed):

    python3 tools/isa_sched.py <mangled-name substring> [library]      e.g.  conv_wide_kernelILi2ELi6ELb0E

Prints the kernel's instructions in program order, compressed to tokens:
    M<n>   n consecutive v_mfma          v<n>  n consecutive other VALU instructions
    DSR / DSW   LDS read / write         GL / GS / GA   global load / store / atomic      DMA  global_load_lds (LDS-DMA)
    W[vm(a) lg(b)]   s_waitcnt with its counts               BAR  s_barrier      BR  branch      n  s_nop      SCR  scratch access
Two patterns this makes visible at a glance: `GL ... W[vm(0)]` a few tokens apart inside a loop whose
loads sit in branches (the compiler cannot count loads in flight behind a branch: every wait becomes vmcnt(0), i.e. a wait for
the loads just issued for the NEXT step), and `DSR W[lg(0)] M<n>` with nothing between (an LDS round trip in front of every MFMA
group).  `SCR` inside a loop is a spilled value; its reload also counts on vmcnt.
The full listing of the kernel is left in /tmp/isa/cur.s."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_hazards as H  # noqa: E402


def summarize(body):
    seq, cnt, val = [], 0, 0
    for line in body:
        line = line.strip()
        if not line:
            continue
        parts = line.split("//")[0].split()
        if not parts:
            continue
        o = parts[0]
        if o.startswith("v_mfma"):
            if val:
                seq.append(f"v{val}")
                val = 0
            cnt += 1
            continue
        if cnt:
            seq.append(f"M{cnt}")
            cnt = 0
        if o.startswith("v_"):
            val += 1
            continue
        if val:
            seq.append(f"v{val}")
            val = 0
        if o == "s_waitcnt":
            seq.append("W[" + " ".join(parts[1:]).replace("vmcnt", "vm").replace("lgkmcnt", "lg") + "]")
        elif o.startswith("ds_read"):
            seq.append("DSR")
        elif o.startswith("ds_write"):
            seq.append("DSW")
        elif o.startswith("global_load_lds"):
            seq.append("DMA")
        elif o.startswith("global_load") or o.startswith("flat_load"):
            seq.append("GL")
        elif o.startswith("global_store"):
            seq.append("GS")
        elif o.startswith("global_atomic"):
            seq.append("GA")
        elif o.startswith("s_cbranch") or o == "s_branch":
            seq.append("BR")
        elif o == "s_barrier":
            seq.append("BAR")
        elif o == "s_nop":
            seq.append("n")
        elif o.startswith("scratch"):
            seq.append("SCR")
    if cnt:
        seq.append(f"M{cnt}")
    if val:
        seq.append(f"v{val}")
    out, prev, k = [], None, 0
    for x in seq:
        if x == prev:
            k += 1
        else:
            if prev is not None:
                out.append(prev + (f"x{k}" if k > 1 else ""))
            prev, k = x, 1
    if prev is not None:
        out.append(prev + (f"x{k}" if k > 1 else ""))
    return out


def main():
    pat = sys.argv[1]
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "idee_amd", "libidee_hip.so")
    os.makedirs("/tmp/isa", exist_ok=True)
    for elf in H.code_objects(lib, "/tmp/isa"):
        text = H.disassemble(elf)
        m = re.search(r"\n[0-9a-f]+ <(_Z[^>]*" + re.escape(pat) + r"[^>]*)>:\n", text)
        if not m:
            continue
        start = m.end()
        end = text.find("\n\n", start)
        body = text[start:end].split("\n")
        with open("/tmp/isa/cur.s", "w") as f:
            f.write("\n".join(body))
        print(m.group(1), len(body), "instructions")
        print(" ".join(summarize(body)))
        return 0
    print("no kernel matches", pat)
    return 1


if __name__ == "__main__":
    sys.exit(main())
