O=gpurun_out/r2ac; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -2
done
