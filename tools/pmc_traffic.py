"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of bench.py into per-kernel-family HBM bytes per launch.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): the counters are in KiB, FETCH_SIZE reports half of a
wide coalesced streaming read on gfx950 and is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, glob, json, re, sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def family(name):
    """kernel symbol -> the tag bench.py / vk_prof use for that kernel family (forward and data-gradient launches share one)"""
    m = re.search(r"conv3x3_col_kernel<vk::(\w+), (\d+), (\d+), (\d+), (\d+), \w+, \d+(?:, (\d+))?>", name)
    if m:
        t, th, bn, wm, wn, stride = m.groups()
        return f"col_{'f32' if t == 'float' else '16b'}_t{th}_bn{bn}_w{int(wm) * int(wn)}" + ("_s2" if stride == "2" else "")
    m = re.search(r"conv3x3_colq_kernel<vk::(\w+), (\d+), (\d+), (\d+), (\d+)>", name)      # the software-pipelined form
    if m:
        t, th, bn, wm, wn = m.groups()
        return f"colq_{'f32' if t == 'float' else '16b'}_t{th}_bn{bn}_w{int(wm) * int(wn)}"
    m = re.search(r"conv3x3_cols_kernel<vk::(\w+), (\d+), (\d+), (\d+), (\d+)>", name)      # the staggered form (opt-in)
    if m:
        t, th, bn, wm, wn = m.groups()
        return f"cols_{'f32' if t == 'float' else '16b'}_t{th}_bn{bn}_w{int(wm) * int(wn)}"
    m = re.search(r"conv3x3_stream_kernel<vk::(\w+), (\d+), (\d+), (\w+), (\d+)>", name)    # streaming small-channel kernel (fwd + dgrad share a symbol per mode)
    if m:
        t, cin, tc, up, mode = m.groups()
        return f"stream_16b_c{cin}{'up' if up == 'true' else ''}_k{int(tc) * 16}" + ("_dgrad" if mode != "0" else "")
    m = re.search(r"conv3x3_colp_kernel<vk::(\w+), (\d+), (\d+), (\d+), (\d+)", name)      # the persistent form runs under the plain tag
    if m:
        t, th, bn, wm, wn = m.groups()
        return f"col_{'f32' if t == 'float' else '16b'}_t{th}_bn{bn}_w{int(wm) * int(wn)}"
    m = re.search(r"conv3x3_s2dg_kernel<vk::(\w+), \d+, (\d+)", name)
    if m:
        return f"s2dg_{'f32' if m.group(1) == 'float' else '16b'}_bn{m.group(2)}"
    m = re.search(r"wgrad_halo_kernel<vk::(\w+), (\d+), (\d+), (\w+), (\w+)(?:, (\d+))?>", name)
    if m:
        t, kt, ct, ws, ts, stride = m.groups()
        return f"wgrad_halo_{'f32' if t == 'float' else '16b'}_{kt}x{ct}{'ts' if ts == 'true' else ''}" + ("_s2" if stride == "2" else "")
    if "stem7x7_kernel" in name:
        return "stem_tile_16b"
    m = re.search(r"vk::(\w+)", name)
    return m.group(1) if m else name[:40]


def collect(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[family(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    res = {"_note": "HBM bytes per launch (mean over every launch of the kernel symbol in `bench.py --steps 2 --warmup 1`): rocprofv3 "
                    "--pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; counters are KiB; FETCH_SIZE doubled for gfx950 "
                    "(MI355X_MICROARCH.md, HBM section)."}
    from bench import kernel_source_hash      # bench.py reports `traffic` only while this stamp matches the tree it runs on
    res["_src_sha256"] = kernel_source_hash()
    for k in sorted(set(fe) | set(wr)):
        f = 2.0 * 1024.0 * sum(fe.get(k, [0])) / max(1, len(fe.get(k, [])))
        w = 1024.0 * sum(wr.get(k, [0])) / max(1, len(wr.get(k, [])))
        res[k] = {"fetch_bytes": round(f), "write_bytes": round(w), "launches": len(fe.get(k, []))}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:36s} fetch {v['fetch_bytes'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB  n={v['launches']}")


if __name__ == "__main__":
    main()
