for i in 1 2; do
 echo "--- default"; python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts']['aekl_gan_train_step']['ms_per_step'], d['parts']['pixel_dm_train_step']['ms_per_step'])"
 echo "--- deterministic"; EEGLDM_DETERMINISTIC=1 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parts']['aekl_gan_train_step']['ms_per_step'], d['parts']['pixel_dm_train_step']['ms_per_step'])"
done
