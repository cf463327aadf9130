O=$PWD/gpurun_out/r2_place; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
$R/tools/placement_pmc 40 > $O/pmc_plain.txt 2>&1; cat $O/pmc_plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pmc_stats -- $R/tools/placement_pmc 40 > $O/pmc_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $O/pmc_tlb -- $R/tools/placement_pmc 40 > $O/pmc_tlb.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/pmc_ea -- $R/tools/placement_pmc 40 > $O/pmc_ea.txt 2>&1
rocprofv3 --kernel-trace --pmc TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum --output-format csv -d $O/pmc_tlb2 -- $R/tools/placement_pmc 40 > $O/pmc_tlb2.txt 2>&1
cd $R
for d in pmc_tlb pmc_ea pmc_tlb2; do echo "== $d: $(cat $O/$d.txt | grep slab)"; python tools/pmc_by_kernel.py $O/$d fill_tag; done
grep fill_tag $O/pmc_stats/*/*kernel_stats.csv 2>/dev/null | cut -c1-200
# keep only the small summaries for the merge-back
find $O -name "*.csv" -size +2M -delete
