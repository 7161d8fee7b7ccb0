#!/bin/bash
# reference-API leg (elm_register on pageable host buffers) with the non-temporal staging copy and with plain memcpy
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export ELM_STAGE_MEMCPY=1; else unset ELM_STAGE_MEMCPY; fi
  python bench.py --no-cpu --batch 256 --steps 2 --warmup 1 > /tmp/r.json 2>/tmp/r.err || tail -3 /tmp/r.err
  python -c "
import json; r=json.load(open('/tmp/r.json'))['reference_api']; print('memcpy' if $v else 'nt-copy', round(r['registrations_per_s']), r['ms_per_call_median'], 'pinned', r['page_locked_source']['ms_per_call_median'])"
done
