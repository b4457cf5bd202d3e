"""Timeline of one frame from a rocprofv3 --kernel-trace csv (development aid): per-kernel busy time, idle gaps."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('::')[-1][:28]))
rows.sort()
# frames are separated by the longest idle gaps; take the last frame
gaps = sorted(((rows[i + 1][0] - max(x[1] for x in rows[max(0, i - 8):i + 1]), i) for i in range(len(rows) - 1)), reverse=True)
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cuts = sorted(i for _, i in gaps[:nframes - 1])
frame = rows[cuts[-1] + 1:] if cuts else rows
t0, t1 = frame[0][0], max(x[1] for x in frame)
busy = collections.defaultdict(float); cnt = collections.Counter()
for s, e, k in frame: busy[k] += (e - s) / 1e6; cnt[k] += 1
# union of busy intervals
u = 0; cur_s, cur_e = frame[0][0], frame[0][1]
for s, e, k in frame[1:]:
    if s > cur_e: u += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
u += cur_e - cur_s
print('frame wall %.2f ms, device busy (union) %.2f ms, idle %.2f ms, kernels %d' % ((t1 - t0) / 1e6, u / 1e6, (t1 - t0 - u) / 1e6, len(frame)))
for k in sorted(busy, key=busy.get, reverse=True): print('  %-28s n=%4d total %.2f ms avg %.3f ms' % (k, cnt[k], busy[k], busy[k] / cnt[k]))
ext = [(s, e) for s, e, k in frame if 'extend<false' in k]
print('  extend per round (ms):', ' '.join('%.2f' % ((e - s) / 1e6) for s, e in ext[:40]), '...', ' '.join('%.2f' % ((e - s) / 1e6) for s, e in ext[40:][-14:]))
seq = [x for x in frame if 'stream_' in x[2]]
g = [(seq[i + 1][0] - seq[i][1]) / 1e3 for i in range(len(seq) - 1)]
if g: print('  gaps between consecutive pipeline kernels (us): mean %.1f  max %.1f  sum %.2f ms' % (sum(g) / len(g), max(g), sum(g) / 1e3))
sh = [(s, e) for s, e, k in frame if 'shade<false' in k]
print('  shade per round (ms):', ' '.join('%.2f' % ((e - s) / 1e6) for s, e in sh[:40]), '...', ' '.join('%.2f' % ((e - s) / 1e6) for s, e in sh[40:][-14:]))
