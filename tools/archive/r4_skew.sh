O=gpurun_out/r4n; mkdir -p $O
bash tools/box_probe.sh $O/box.json > /dev/null 2>&1
A="PGX_XCD_SKEW=0:.125:.25:.375:.5:.625:.75:.875"; B="PGX_XCD_SKEW=0:.5:0:.5:0:.5:0:.5"; C="PGX_XCD_SKEW=0:.5:.25:.75:.125:.625:.375:.875"; D="PGX_XCD_SKEW=.03:.41:.77:.19:.58:.91:.33:.66"
AB_PLAIN_BUFFERS=1 python tools/ab_inproc.py cfg2 "" "$A" "$B" "$C" "$D" "" > $O/skew_plain_cfg2.txt 2>&1
python tools/ab_inproc.py cfg2 "" "$A" "$C" "$D" "" > $O/skew_zone_cfg2.txt 2>&1
AB_PLAIN_BUFFERS=1 python tools/ab_inproc.py cfg4 "" "$A" "$C" "$D" > $O/skew_plain_cfg4.txt 2>&1
tail -7 $O/skew_plain_cfg2.txt; tail -6 $O/skew_zone_cfg2.txt; tail -5 $O/skew_plain_cfg4.txt
