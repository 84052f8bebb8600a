#!/bin/bash
# builds libawt variants that differ only in attention_f8.hip compile-time switches -> mlx8-ws-audio-transformer_amd/variants/
set -e
cd "$(dirname "$0")/.."
P=mlx8-ws-audio-transformer_amd; mkdir -p $P/variants; rm -f $P/variants/libawt_v*.so
python -m mlx8_ws_audio_transformer_amd.build > /dev/null
i=0
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc $flags -c $P/csrc/attention_f8.hip -o $P/variants/attention_f8_v$i.o 2>/dev/null
  objs=$(ls $P/csrc/_obj/*.o | grep -v attention_f8.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/variants/libawt_v$i.so $objs $P/variants/attention_f8_v$i.o -ldl
  echo "v$i: $flags"
  i=$((i+1))
done
