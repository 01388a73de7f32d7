cd $GRAFT_REPO_ROOT
for cfg in "- -" "1 4" "1 8" "1 16" "2 8" "2 16"; do set -- $cfg; L=$1; W=$2; if [ $L = - ]; then env FRAMES=4096 python tools/time_parse.py; else env FRAMES=4096 NVH_PARSE_LANES=$L NVH_PARSE_WAVES=$W python tools/time_parse.py; fi; done
for cfg in "- -" "4 16" "8 16" "2 16" "8 8" "16 8"; do set -- $cfg; L=$1; W=$2; if [ $L = - ]; then env FRAMES=32768 python tools/time_parse.py; else env FRAMES=32768 NVH_PARSE_LANES=$L NVH_PARSE_WAVES=$W python tools/time_parse.py; fi; done
