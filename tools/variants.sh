#!/bin/bash
# Build kernel variants (-D flags) in scratch copies on the GPU box and time each with bench.py: `value` (four launch sequences in
# flight), `serial` (one at a time) and the per-kernel durations, all from ONE run per variant on ONE box.
# Usage: tools/variants.sh "name1:-DFOO=1 -DBAR=2" "name2:..." ...      (timing experiments; outputs may be wrong)
#   VARIANT_PARITY="name ..."   run the -m gpu suite in those variants' copies first (a variant meant to ship must pass it)
#   VARIANT_PATHS="C2 C3-LUT"   the other single-GPU paths through tools/bench_paths.py as well
#   BENCH_ARGS="..."            extra bench.py arguments
SRC=$PWD
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1', 'in-flight', d['value'], d['roofline']['frac'], '| serial', d.get('serial',{}).get('frames_per_s'), d.get('serial',{}).get('roofline_frac'),
      '|', [(k['name'], k['ms_per_frame']) for k in d['roofline']['kernels']], '| lone', d.get('latency_mode',{}).get('frames_per_s'),
      'one-per-call in flight', d.get('latency_mode',{}).get('pipelined_calls',{}).get('frames_per_s'))"; }
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  # the scratch copy is keyed on the sources AND the flags (a stale copy under the same name is never timed); VARIANT_REUSE=1 keeps
  # an existing copy of the same key (repeat timings of one build), a failed build removes its directory
  key=$( (cat $SRC/stitchingvideo_amd/csrc/*.hip $SRC/stitchingvideo_amd/csrc/*.[ch]pp $SRC/include/*.h $SRC/bench.py; echo "$flags") | sha256sum | cut -c1-12)
  W=/tmp/mst_var_${name}_$key
  [ -z "$VARIANT_REUSE" ] && rm -rf $W
  if [ ! -d $W ]; then
    mkdir -p $W; cp -r $SRC/stitchingvideo_amd $SRC/include $SRC/tools $SRC/bench.py $SRC/tests $SRC/oracle $SRC/__graft_entry__.py $W/ 2>/dev/null
    (cd $W/stitchingvideo_amd/csrc && rm -rf build && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags" > $W/build.log 2>&1) || { echo "$name: build failed"; tail -5 $W/build.log; rm -rf $W; continue; }
    case " $VARIANT_PARITY " in *" $name "*) (cd $W && python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2 | sed "s/^/$name parity: /");; esac
  fi
  # VARIANT_LATENCY=1: with the one-frame-set legs (latency_mode) -- the host-buffer and ingest legs ride along, so it takes longer
  LAT="--no-latency-leg"; [ -n "$VARIANT_LATENCY" ] && LAT=""
  (cd $W && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs $LAT --min-time 0.4 $BENCH_ARGS 2>/dev/null | show "$name")
  [ -n "$VARIANT_PATHS" ] && (cd $W && python3 tools/bench_paths.py $VARIANT_PATHS --no-latency-leg 2>/dev/null | python3 -c "
import sys,json
print('$name', ' '.join('%s %.4f' % (d['name'], d.get('roofline_frac', -1)) for d in map(json.loads, filter(lambda l: l.startswith('{'), sys.stdin))))")
done
