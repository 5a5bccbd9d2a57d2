"""Instruction census of one kernel from hipcc's device assembly (hipcc --cuda-device-only -S).

usage: isa_census.py file.s kernel_substring [first_label last_label]
Splits the kernel into basic blocks (labels) and counts instructions per class and block; with two labels only the blocks
from the first label up to (not including) the second are listed -- e.g. the loop body of k_accumulate."""
import re, sys, collections

CLASSES = [
    ("mad_u64_u32", re.compile(r"^v_mad_u64_u32")),
    ("mul_lo/hi_u32", re.compile(r"^v_mul_(lo|hi)_u32")),
    ("shift64", re.compile(r"^v_(lshrrev|lshlrev|ashrrev)_b64")),
    ("add64", re.compile(r"^v_lshl_add_u64|^v_add_co|^v_addc_co|^v_sub_co|^v_subb_co|^v_subrev_co|^v_subbrev_co")),
    ("and/or/xor", re.compile(r"^v_(and|or|xor|or3|and_or|bfe|bfi|not|lshl_or)_")),
    ("shift32", re.compile(r"^v_(lshrrev|lshlrev|ashrrev|alignbit)_b32")),
    ("add/sub32", re.compile(r"^v_(add|sub|subrev|add3|lshl_add|add_lshl|mad_u32_u24|mad_i32_i24)_(u32|nc_u32|i32)|^v_add3_u32|^v_lshl_add_u32|^v_add_lshl_u32")),
    ("cndmask", re.compile(r"^v_cndmask")),
    ("cmp", re.compile(r"^v_cmp")),
    ("mov", re.compile(r"^v_mov|^v_accvgpr|^v_readfirstlane|^v_readlane|^v_writelane")),
    ("valu_other", re.compile(r"^v_")),
    ("vmem", re.compile(r"^(global|buffer|flat|scratch)_")),
    ("lds", re.compile(r"^ds_")),
    ("waitcnt/nop", re.compile(r"^s_(waitcnt|nop|sleep|barrier)")),
    ("branch", re.compile(r"^s_(cbranch|branch|endpgm|setpc|call)")),
    ("salu", re.compile(r"^s_")),
]
VALU = {"mad_u64_u32", "mul_lo/hi_u32", "shift64", "add64", "and/or/xor", "shift32", "add/sub32", "cndmask", "cmp", "mov", "valu_other"}

def classify(m):
    for name, rx in CLASSES:
        if rx.match(m):
            return name
    return "other"

def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().split(":")[0].startswith("_Z") and ":" in l)
    blocks = collections.OrderedDict()
    cur = "entry"; blocks[cur] = collections.Counter()
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1); blocks[cur] = collections.Counter(); continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        mn = s.split()[0]
        blocks[cur][classify(mn)] += 1
    names = list(blocks)
    if len(sys.argv) >= 5:
        a, b = names.index(sys.argv[3]), names.index(sys.argv[4])
        names = names[a:b]
    cols = [c for c, _ in CLASSES]
    print(f"{'block':12s} {'total':>6s} {'VALU':>6s} " + " ".join(f"{c[:11]:>11s}" for c in cols))
    tot = collections.Counter()
    for n in names:
        c = blocks[n]; tot.update(c)
        print(f"{n:12s} {sum(c.values()):6d} {sum(v for k, v in c.items() if k in VALU):6d} " + " ".join(f"{c.get(k, 0):11d}" for k in cols))
    print(f"{'sum':12s} {sum(tot.values()):6d} {sum(v for k, v in tot.items() if k in VALU):6d} " + " ".join(f"{tot.get(k, 0):11d}" for k in cols))

main()
