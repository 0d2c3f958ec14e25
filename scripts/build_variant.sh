# builds a variant of the library for scripts/gpu_ab*.sh: bash scripts/build_variant.sh NAME "<extra hipcc flags>"
#   -> sphfluidsimulation_amd/lib/variants/NAME.so (the default library is rebuilt without the flags afterwards)
set -e
mkdir -p sphfluidsimulation_amd/lib/variants
SPH_HIPCC_EXTRA="$2" python -m sphfluidsimulation_amd.build --force > /dev/null
cp sphfluidsimulation_amd/lib/libsph_hip.so sphfluidsimulation_amd/lib/variants/$1.so
echo "built variants/$1.so [$2]"
