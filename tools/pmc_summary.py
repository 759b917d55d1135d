"""Summarises the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) for the dominant
kernel into profiles/<round>/pmc_gather_kernel.json.  gfx950 corrections per MI355X_MICROARCH.md (HBM):
both counters are in KiB; FETCH_SIZE tallies 128-byte requests at 64 B, i.e. reports half the bytes of a
wide (16 B/lane) coalesced read stream -> doubled."""
import csv, glob, json, sys

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]


def per_launch(d, name):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    def last_op(n):         # gather_kernel<..., true>(...): the instance launched for a batch's last op (kernels_gather.hip LASTOP)
        return 'gather_kernel' in n and n[:n.rfind('(')].rstrip().endswith('true>')
    rows = [r for r in csv.DictReader(open(f)) if last_op(r['Kernel_Name']) and r['Counter_Name'] == name]
    mx = max(int(r['Grid_Size']) for r in rows)                   # ... of a full group
    sel = [float(r['Counter_Value']) for r in rows if int(r['Grid_Size']) == mx]
    return sum(sel) / len(sel), len(sel), mx


f, nf, grid = per_launch(fetch_dir, 'FETCH_SIZE')
w, nw, _ = per_launch(write_dir, 'WRITE_SIZE')
bench = json.loads(open(fetch_dir + '/bench.json').read().strip().splitlines()[-1])
rows = bench['roofline']['rows_per_launch']
res = {"kernel": "lg::gather_kernel<float4, ROWS, 4, false, true> (the last hop's gather of a full lane group)", "grid_threads": grid,
       "launches_sampled": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
       "FETCH_SIZE_KiB_raw_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
       "read_bytes_per_launch_corrected": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
       "traffic_bytes_per_launch": (2 * f + w) * 1024,
       "rows_per_launch": rows, "algorithmic_bytes_per_launch": rows * bench['roofline']['bytes_per_row'],
       "traffic_over_algorithmic": (2 * f + w) * 1024 / (rows * bench['roofline']['bytes_per_row']),
       "config": bench['config']['workload'], "batches_per_launch_group": bench['config']['batches_per_launch_group'],
       "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for 16 B/lane streams); WRITE_SIZE exact; both in KiB"}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
