for d in 0 128 7 135; do echo "dbg=$d"; NR_RP_DBG=$d python tools/rowpanel_ab.py 2>&1 | grep "M= 32768" | grep -v "square " ; done
