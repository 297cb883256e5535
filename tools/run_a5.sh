for w in svgd128 map256 map5; do
python tools/graph_probe.py $w 2>/dev/null
PACOH_NO_GRAPH=1 python tools/graph_probe.py $w 2>/dev/null
PACOH_MLP_PATH=mfma python tools/graph_probe.py $w 2>/dev/null
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_probe.py $w 2>/dev/null
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python tools/graph_probe.py $w 2>/dev/null
done
