# mapping knobs of the grid sweep kernel at 257^3 (V-cycle top level): bash tools/grid257_knobs.sh
export PMG_VC_ONLY=257
for v in "" "PMG_GRID_TAIL=0" "PMG_GRID_TAIL=0 PMG_GRID_PACKED=1" "PMG_GRID_TAIL=0 PMG_GRID_PACKED=0" "PMG_GRID_BANDED=0" "PMG_GRID_TAIL=0 PMG_GRID_BANDED=0"; do
  echo "[$v] $(env $v python tools/vcyclebench.py 2>&1 | grep V-cycle)"
done
