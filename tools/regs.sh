#!/bin/bash
# Register / scratch report of one HIP source (run after every kernel edit: DESIGN findings 9, 28).
# usage: tools/regs.sh speech-separation_amd/csrc/lstm_rs.hip [extra hipcc flags]
src=$1; shift
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value "$@" \
      -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
  | awk '/Function Name/ {if (line) print line; line=$3; next} {gsub(/^ +/,""); line=line " | " $0} END {print line}' \
  | while read -r l; do n=$(echo "$l" | cut -d' ' -f1 | c++filt | cut -c1-70); echo "$n |${l#* |}"; done
