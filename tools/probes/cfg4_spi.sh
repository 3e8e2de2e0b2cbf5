# BASELINE config 4: which resource keeps further workgroups off a CU while the bisecting sweep runs?  SPI resource-allocation counters.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/pmc4_r03_spi
mkdir -p $D
rocprofv3 --list-avail > $D/avail.txt 2>&1
grep -o "SPI_RA_[A-Z_0-9]*" $D/avail.txt | sort -u > $D/spi_names.txt
cat $D/spi_names.txt | tr '\n' ' '
echo
for grp in "SPI_RA_REQ_NO_ALLOC SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_TMP_STALL_CSN" "SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN" "SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN SPI_CSN_BUSY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $D/$tag -- python3 tools/probes/cfg4_modes.py 4 20 > $D/$tag.log 2>&1
  python3 - $D/$tag <<'PY'
import csv,glob,collections,os,sys
fs=sorted(glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'), key=os.path.getmtime)
if not fs:
    print('no counters in', sys.argv[1], open(sys.argv[1]+'.log').read()[-400:]); sys.exit(0)
rows=[r for r in csv.DictReader(open(fs[-1])) if r['Kernel_Name'].startswith('gfh_k_sweep')]
ids=sorted({int(r['Dispatch_Id']) for r in rows})[-20:]
agg=collections.defaultdict(list)
for r in rows:
    if int(r['Dispatch_Id']) in ids: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: '%.5g'%(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
done
