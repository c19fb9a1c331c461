timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 2>&1 | grep -E "^[0-9.]+s (call|setup)" | head -30
