"""Runs a command and samples the GPU's shader clock, memory clock, power and temperature from sysfs while it runs.

    python tools/clock_watch.py [--period 0.02] -- python bench.py ...

Prints the command's stdout unchanged and, on stderr, one JSON line: the clocks the timed work actually ran at (a figure in
cycles, such as the VALU issue bound of bench.py's `roofline.issue`, is only as good as the clock it assumes).  The sampler never
touches the HIP runtime; the command runs as a child process."""
import glob
import json
import subprocess
import sys
import threading
import time


def sources():
    """Per card (a box may show several; only the one the job runs on gets busy): sysfs files of clocks, power, temperature."""
    cards = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        card = hw.split("/")[4]
        out = cards.setdefault(card, {})
        for key, names in (("sclk_hz", ["freq1_input"]), ("mclk_hz", ["freq2_input"]),
                           ("power_uw", ["power1_average", "power1_input"]), ("temp_mc", ["temp1_input", "temp2_input"])):
            for n in names:
                p = "%s/%s" % (hw, n)
                try:
                    int(open(p).read().strip())
                    out.setdefault(key, p)
                except (OSError, ValueError):
                    pass
        p = "/sys/class/drm/%s/device/gpu_busy_percent" % card
        try:
            int(open(p).read().strip())
            out["busy_pct"] = p
        except (OSError, ValueError):
            pass
    return cards


def main():
    argv = sys.argv[1:]
    period = 0.02
    if argv and argv[0] == "--period":
        period = float(argv[1])
        argv = argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    cards = sources()
    samples = {c: {k: [] for k in src} for c, src in cards.items()}
    stop = threading.Event()

    def sample():
        while not stop.is_set():
            for c, src in cards.items():
                for k, p in src.items():
                    try:
                        samples[c][k].append(int(open(p).read().strip()))
                    except (OSError, ValueError):
                        samples[c][k].append(-1)
            time.sleep(period)
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    rc = subprocess.call(argv)
    stop.set()
    th.join()
    # the card the command ran on: the one that was busiest
    def mean_busy(c):
        b = [x for x in samples[c].get("busy_pct", []) if x >= 0]
        return sum(b) / len(b) if b else -1.0
    rep = {"cards": sorted(cards), "mean_busy": {c: round(mean_busy(c), 1) for c in cards}}
    if cards:
        card = max(cards, key=mean_busy)
        rep["card"] = card
        busy = samples[card].get("busy_pct", [])
        for k, v in samples[card].items():
            # "under load": the samples taken while the GPU reported itself busy, when that signal exists
            hot = [x for x, b in zip(v, busy) if b >= 90 and x >= 0] if busy and len(busy) == len(v) else [x for x in v if x >= 0]
            s = sorted(hot or [x for x in v if x >= 0])
            if s:
                rep[k] = {"min": s[0], "p10": s[len(s) // 10], "median": s[len(s) // 2], "p90": s[(len(s) * 9) // 10], "max": s[-1], "n": len(s)}
    print(json.dumps(rep), file=sys.stderr)
    sys.exit(rc)


if __name__ == "__main__":
    main()
