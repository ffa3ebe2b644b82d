"""What hipcc made of the loads: per kernel of reviews4rec_amd/csrc/*.hip (compiled to gfx950 ISA here, no GPU needed)
  - VGPRs, scratch bytes (small arrays indexed by a run-time trip count land in scratch: DESIGN 4.2),
  - vector loads that are WAITED FOR AT ONCE (`s_waitcnt vmcnt(0)` within four instructions and no other load in
    between): a load inside a branch with its use, a load behind a uniform `if`, a pointer fetched from the argument
    segment -- in a prologue or a loop each one is a dependent memory round trip,
  - scalar registers spilled into vector lanes and the v_readlane / v_writelane instructions that serve them (in an
    arithmetic-bound loop they are issue slots: DESIGN 4.7).
`python tools/isa_scan.py [unit ...]` (default: every translation unit); prints the kernels with three or more."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'reviews4rec_amd', 'csrc')
units = sys.argv[1:] or sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, '*.hip')))
for unit in units:
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', tmp.name,
                        os.path.join(CSRC, unit)], check=True, stderr=subprocess.DEVNULL)
        s = open(tmp.name).read()
    for m in re.finditer(r'\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel', s, re.S):
        name, meta = m.group(1), m.group(2)
        i = s.index(name + ':')
        ins = [l.strip() for l in s[i:s.find('.Lfunc_end', i)].split('\n')
               if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
        hits = []
        for n, l in enumerate(ins):
            if l.startswith(('global_load', 'buffer_load')):
                nxt = ins[n + 1:n + 5]
                if any(x.startswith('s_waitcnt vmcnt(0)') for x in nxt) and not nxt[0].startswith(('global_load', 'buffer_load')):
                    hits.append(n)
        vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', meta).group(1)
        sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', meta).group(1)
        spills = sum('scratch_' in l for l in ins)
        md = re.search(r'\.name:\s+' + re.escape(name) + r'\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)', s)
        sspill = int(md.group(1)) if md else 0             # scalar registers spilled into vector lanes: v_readlane / v_writelane traffic
        lane = sum(l.startswith(('v_readlane', 'v_writelane')) for l in ins)
        if len(hits) >= 3 or spills or sspill >= 64:
            print('%-22s %-62s %5d instr  %3s VGPRs  scratch %3s B (%d ops)  scalar spills %3d (%d lane ops)  loads waited for at once: %2d  %s'
                  % (unit, name[8:70], len(ins), vg, sc, spills, sspill, lane, len(hits), hits[:8]))
