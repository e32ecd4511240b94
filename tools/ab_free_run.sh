#!/bin/bash
# developer A/B (GPU box): the size below which pt_render's batches stop advancing in lockstep (MI355PT_FREE_RUN_BELOW, paths per batch): rank 0 of an 8-way sharded C3 frame | the full frame
for fr in 0 65536 262144 1048576 4194304 16777216 0 1048576; do
  echo "free run below $fr: $(MI355PT_FREE_RUN_BELOW=$fr python tools/rank_profile.py 8 8 2>/dev/null | tail -1 | cut -c1-60) | $(MI355PT_FREE_RUN_BELOW=$fr python tools/rank_profile.py 1 5 2>/dev/null | tail -1 | cut -c1-60)"
done
