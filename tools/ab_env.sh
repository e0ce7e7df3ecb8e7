# bench.py --no-cpu for a list of "NAME=VALUE,NAME=VALUE" settings (':' for none); output tagged by index
i=0
for setting in "$@"; do
  envs=$(echo "$setting" | tr ',' ' ')
  if [ "$setting" = ":" ]; then envs=""; fi
  env $envs python bench.py --no-cpu 2>/dev/null > gpurun_out/ab_$i.json
  i=$((i+1))
done
