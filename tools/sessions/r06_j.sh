cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06j; mkdir -p $O
python tools/gpu_wave_company.py 2>&1 | grep -v amdgpu.ids | tee $O/wave_company.txt
bash tools/scale_check.sh > $O/scale_check_n1.txt 2>&1; echo "scale_check rc=$?"; tail -25 $O/scale_check_n1.txt | cut -c1-250
python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
